#!/bin/bash
# kernel-trace stats of the single-stream bench (gpurun): bash tools/profile_stats.sh <cfg>
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
CFG=${1:-c2}
OUT=gpurun_out/stats_$CFG
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s1 -- python3 bench.py --config $CFG --steps 20 --warmup 5 --no-cpu-baseline --no-side-legs --streams 1 > $OUT/s1_line.json 2> $OUT/s1.err
python3 tools/stats_md.py $OUT/s1/*/*_kernel_stats.csv "x" "y" 14
