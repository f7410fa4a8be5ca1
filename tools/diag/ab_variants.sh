#!/bin/bash
# A/B of tuning builds on the GPU box: kernel stats (single stream) per library variant.
#   bash tools/diag/ab_variants.sh <cfg> <lib or "default"> ...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
CFG=$1; shift
for V in "$@"; do
  if [ "$V" = "default" ]; then unset GROUPNET_HIP_LIB; else export GROUPNET_HIP_LIB=$GRAFT_REPO_ROOT/tools/diag/$V; fi
  OUT=gpurun_out/ab_$V; rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s1 -- python3 bench.py --config $CFG --steps 20 --warmup 5 --no-cpu-baseline --no-side-legs --streams 1 > $OUT/line.json 2> $OUT/err
  echo "== $V"; python3 tools/stats_md.py $OUT/s1/*/*_kernel_stats.csv x y 12 | grep -E "_x_kernel|_xs_kernel|node_stage|node2edge|affinity|rb2" | cut -c1-80
  python3 -c "
import json; l=json.loads(open('$OUT/line.json').read().strip().splitlines()[-1]); print('   value', l['value'], 'ms', l['ms_per_step'])"
done
