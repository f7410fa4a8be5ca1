#!/bin/bash
# throughput of the default bench for several stream counts (and the affinity-tail form), one line each
cd "$GRAFT_REPO_ROOT"
for S in 2 3 4 6 8; do
  python3 bench.py --config c2 --no-cpu-baseline --no-side-legs --streams $S > gpurun_out/sweep_$S.json 2>gpurun_out/sweep_$S.err
  python3 -c "
import json; l=json.loads(open('gpurun_out/sweep_$S.json').read().strip().splitlines()[-1]); print('streams', $S, 'value', l['value'], 'ms', l['ms_per_step'], l.get('ms_per_step_min_max'))"
done
