#!/bin/bash
# the two PMC passes of the single-stream bench (gpurun): bash tools/profile_pmc.sh <cfg> <tag> [env...]
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
CFG=${1:-c2}; TAG=${2:-x}
OUT=gpurun_out/pmc_${CFG}_$TAG
rm -rf $OUT && mkdir -p $OUT
COMMON="--config $CFG --steps 20 --warmup 5 --no-cpu-baseline --no-side-legs --streams 1"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py $COMMON > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py $COMMON > /dev/null 2> $OUT/pmc_write.err
python3 tools/pmc_traffic.py $OUT/pmc_fetch $OUT/pmc_write "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE --output-format csv -- python3 bench.py $COMMON" > $OUT/traffic.json
python3 - <<PY
import json
d=json.load(open("$OUT/traffic.json"))
tot=0
for k,v in d["kernels"].items():
    if v["launches"]>=40 and "rocclr" not in k and "spin" not in k:
        print(k, v["FETCH_SIZE_KB"], v["WRITE_SIZE_KB"], round(v["hbm_bytes"]/1e6,1)); tot+=v["hbm_bytes"]
print("$TAG sum MB per forward", round(tot/1e6,1))
PY
