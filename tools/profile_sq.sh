#!/bin/bash
# SQ wave-cycle breakdown per kernel (gpurun): bash tools/profile_sq.sh <cfg>
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
CFG=${1:-c2}
OUT=gpurun_out/sq_$CFG
rm -rf $OUT && mkdir -p $OUT
COMMON="--config $CFG --steps 10 --warmup 3 --no-cpu-baseline --no-side-legs --streams 1"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/p1 -- python3 bench.py $COMMON > /dev/null 2> $OUT/p1.err
python3 - <<PY
import csv, glob, collections, re
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob("$OUT/p1/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); name = re.sub(r"^void ", "", name)[:40]
        rows[name + "@" + r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in rows.items():
    n = len(c.get("SQ_WAVE_CYCLES", []))
    if n < 8 or "spin" in k or "rocclr" in k: continue
    m = {a: sum(v) / len(v) for a, v in c.items()}
    wc = m["SQ_WAVE_CYCLES"]
    print(f"{k:55s} n={n:4d} wave_cycles(quad)={wc:12.0f}  WAIT_ANY {m['SQ_WAIT_ANY']/wc:5.2f}  WAIT_INST_ANY {m['SQ_WAIT_INST_ANY']/wc:5.2f}  ACTIVE_INST_ANY {m['SQ_ACTIVE_INST_ANY']/wc:5.2f}  "
          f"WAIT_INST_LDS {m['SQ_WAIT_INST_LDS']/wc:5.2f}  ACT_VALU {m['SQ_ACTIVE_INST_VALU']/wc:5.2f}  ACT_LDS {m['SQ_ACTIVE_INST_LDS']/wc:5.2f}  MFMA_BUSY(cyc) {m['SQ_VALU_MFMA_BUSY_CYCLES']:12.0f}")
PY
