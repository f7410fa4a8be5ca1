#!/bin/bash
# A/B of environment knobs on the GPU box: kernel stats (single stream) + overlapped throughput per setting.
#   bash tools/diag/ab_env.sh <cfg> "VAR=val VAR2=val" "..." ...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
CFG=$1; shift
i=0
for V in "$@"; do
  i=$((i+1)); OUT=gpurun_out/abenv_$i; rm -rf $OUT; mkdir -p $OUT
  ( export $V; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s1 -- python3 bench.py --config $CFG --steps 20 --warmup 5 --no-cpu-baseline --no-side-legs --streams 1 > $OUT/line.json 2> $OUT/err
    python3 bench.py --config $CFG --no-cpu-baseline --no-side-legs > $OUT/line4.json 2>> $OUT/err )
  echo "== $V"; python3 tools/stats_md.py $OUT/s1/*/*_kernel_stats.csv x y 12 | grep -E "_x_kernel|_xs_kernel|node_stage|node2edge|affinity|rb2" | cut -c1-80
  python3 -c "
import json; l=json.loads(open('$OUT/line.json').read().strip().splitlines()[-1]); m=json.loads(open('$OUT/line4.json').read().strip().splitlines()[-1]); print('   single-stream ms', l['ms_per_step'], '| 4-stream value', m['value'], 'ms', m['ms_per_step'])"
done
