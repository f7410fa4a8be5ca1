#!/usr/bin/env python3
"""rocprofv3 PMC passes -> profiles/rNN_pmc_traffic_<cfg>.json (HBM bytes per launch, per kernel).

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py ...
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py ...
    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write "<command>" > profiles/r02_pmc_traffic_c2.json

Collected and corrected as MI355X_MICROARCH.md (HBM section, rocprofv3 PMC slots) prescribes: the two counters in
SEPARATE passes, each with --kernel-trace only; FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports half
of the bytes of a wide coalesced read, so hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.  Keys are
kernel@grid<threads>: a kernel launched with a different grid (another batch size) is another key.
"""
import csv
import glob
import json
import os
import re
import sys


def demangle(name):
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
        m = re.search(r"\d+([a-z][a-z0-9_]*_kernel)", name)
        return m.group(1) if m else name


def short(name):
    name = demangle(name)
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"[<(].*$", "", name)


def collect(d, counter):
    out = {}
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if r.get("Counter_Name") != counter:
                continue
            key = f"{short(r['Kernel_Name'])}@grid{r['Grid_Size']}"
            # one row per dispatch and counter (summed over instances by rocprofv3's csv writer when it lists dimensions)
            did = r["Dispatch_Id"]
            e = out.setdefault(key, {})
            e[did] = e.get(did, 0.0) + float(r["Counter_Value"])
    return {k: (sum(v.values()) / len(v), len(v)) for k, v in out.items()}


def main():
    fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        f, nf = fetch.get(k, (0.0, 0))
        w, nw = write.get(k, (0.0, 0))
        kernels[k] = dict(launches=max(nf, nw), FETCH_SIZE_KB=round(f, 1), WRITE_SIZE_KB=round(w, 1),
                          hbm_bytes=int((2 * f + w) * 1024))
    print(json.dumps(dict(
        note="rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes (MI355X_MICROARCH.md: HBM section and "
             "rocprofv3 PMC slots), each with --kernel-trace only; units KB; per-launch averages; hbm_bytes = "
             "(2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE reads half of a wide coalesced stream: calibrated in "
             "round 1 on agg_gather/agg_scatter at B=4096, whose corrected totals equal the algorithmic byte counts; the "
             "scattered 16-byte operand loads of the MFMA kernels are NOT calibrated).  Keys are kernel@grid (threads).",
        command=sys.argv[3] if len(sys.argv) > 3 else "", kernels=kernels), indent=1))


if __name__ == "__main__":
    main()
