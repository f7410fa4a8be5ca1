#!/bin/bash
# Runs on the GPU box (gpurun): kernel-trace stats (single stream + default) and the two PMC passes of bench.py,
# written under gpurun_out/; summaries are copied into profiles/ afterwards by hand (tools/stats_md.py, pmc_traffic.py).
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
CFG=${1:-c2}
OUT=gpurun_out/prof_$CFG
rm -rf $OUT && mkdir -p $OUT
COMMON="--config $CFG --steps 20 --warmup 5 --no-cpu-baseline --no-train-leg --no-c4-leg"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s1 -- python3 bench.py $COMMON --streams 1 > $OUT/s1_line.json 2> $OUT/s1.err
echo "s1 done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/dflt -- python3 bench.py $COMMON > $OUT/dflt_line.json 2> $OUT/dflt.err
echo "default done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py $COMMON --streams 1 > /dev/null 2> $OUT/pmc_fetch.err
echo "fetch done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py $COMMON --streams 1 > /dev/null 2> $OUT/pmc_write.err
echo "write done"
find $OUT -name "*.csv" | head -20
du -sh $OUT
