#!/usr/bin/env python3
"""rocprofv3 `*_kernel_stats.csv` -> the short markdown table kept under profiles/.

    python tools/stats_md.py <kernel_stats.csv> "<title>" "<command>" > profiles/<name>.md
"""
import csv
import re
import sys


def demangle(name: str) -> str:
    if not name.startswith("_Z"):
        return name
    import shutil, subprocess
    tool = shutil.which("llvm-cxxfilt") or shutil.which("c++filt") or "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"
    try:
        out = subprocess.run([tool, name], capture_output=True, text=True, check=True).stdout.strip()
        if out and not out.startswith("_Z"):
            return out
        raise ValueError(name)
    except Exception:
        m = re.search(r"\d+([a-z][a-z0-9_]*_kernel)I(.*?)E+v", name)      # e.g. bf16 template arguments llvm-cxxfilt does not know
        if m:
            args = m.group(2).replace("Li1EDF16b", "1, __bf16").replace("Li2EDF16b", "2, __bf16").replace("DF16b", "__bf16")
            return f"{m.group(1)}<{args}>"
        return name


def short(name: str) -> str:
    name = demangle(name)
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
