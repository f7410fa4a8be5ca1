#!/usr/bin/env python3
"""rocprofv3 `*_kernel_stats.csv` -> the short markdown table kept under profiles/.

    python tools/stats_md.py <kernel_stats.csv> "<title>" "<command>" > profiles/<name>.md
"""
import csv
import re
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\(.*$", "", name)
    name = re.sub(r"^void ", "", name)
    return name[:70]


def main():
    path, title, cmd = sys.argv[1], sys.argv[2], sys.argv[3]
    rows = list(csv.DictReader(open(path)))
    print(f"# {title}\n\nCommand (on the MI355X box): `{cmd}`\n")
    print("| kernel | calls | avg us | min us | max us | % of GPU time |\n|---|---|---|---|---|---|")
    for r in rows[:int(sys.argv[4]) if len(sys.argv) > 4 else 16]:
        print(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['MinNs']) / 1e3:.1f} | "
              f"{float(r['MaxNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} |")


if __name__ == "__main__":
    main()
