#!/bin/bash
# CPU side: turn gpurun_out/final3 (written by tools/final_r03.sh on the GPU box) into the committed profiles/r03_* files.
set -e
cd "$(dirname "$0")/.."
O=gpurun_out/final3   # (delete the local copy before a new gpurun: files of earlier runs are merged, not replaced)
P=profiles
C="--steps 20 --warmup 5 --no-cpu-baseline --no-side-legs"
for CFG in c2 c4; do
  [ $CFG = c2 ] && D="config 2 (B=512, N=11, scales {2,5,11}, fp32 results)" || D="config 4 (B=1024, N=50, scales {2,4,8,16}, bf16 twins)"
  csv=$(ls -t $O/${CFG}_s1/*/*_kernel_stats.csv | head -1)
  cp $csv $P/r03_${CFG}_streams1_kernel_stats.csv
  python3 tools/stats_md.py $csv "Round 3 — kernel stats, $D, ONE stream" "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --config $CFG $C --streams 1" 14 > $P/r03_${CFG}_streams1_kernel_stats.md
  python3 tools/pmc_traffic.py $O/${CFG}_fetch $O/${CFG}_write "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE --output-format csv -- python3 bench.py --config $CFG $C --streams 1" > $P/r03_pmc_traffic_${CFG}.json
  tail -1 $O/${CFG}_s1_line.json > $P/r03_bench_${CFG}_streams1_line.json
done
csv=$(ls -t $O/c2_dflt/*/*_kernel_stats.csv | head -1)
cp $csv $P/r03_c2_default_kernel_stats.csv
python3 tools/stats_md.py $csv "Round 3 — kernel stats, config 2, default bench (4 graphs round-robin on 4 streams)" "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --config c2 $C" 14 > $P/r03_c2_default_kernel_stats.md
csv=$(ls -t $O/c2_train/*/*_kernel_stats.csv | head -1)
cp $csv $P/r03_train_step_kernel_stats.csv
python3 tools/stats_md.py $csv "Round 3 — kernel stats of the graphed training step (B=512, N=11)" "rocprofv3 --kernel-trace --stats --output-format csv -- python3 tools/train_step_time.py" 18 > $P/r03_train_step_kernel_stats.md
tail -1 $O/bench_c2.json > $P/r03_bench_c2_line.json
tail -1 $O/bench_c4.json > $P/r03_bench_c4_line.json
tail -1 $O/bench_fd.json > $P/r03_bench_force_dist_line.json
{ echo "# SQ counter breakdown, single-stream bench (tools/profile_sq.sh), config 2 then config 4"; cat $O/sq_c2.txt; echo; cat $O/sq_c4.txt; } > $P/r03_sq_breakdown.txt
ls -la $P | grep r02
tail -1 $O/bench_c2_bf16x6.json > $P/r03_bench_c2_bf16x6_line.json
cp $O/stamps_c2.txt $P/r03_stamps_c2.txt
