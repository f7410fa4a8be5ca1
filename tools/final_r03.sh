#!/bin/bash
# Round-3 evidence run (gpurun): tests, bench lines, kernel stats, PMC traffic, SQ breakdown, training step.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/final3
rm -rf $O && mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; tail -1 $O/tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
python bench.py --gpus 1 --steps 200 --warmup 20 > $O/bench_c2.json 2> $O/bench_c2.err && echo bench_c2 ok
python bench.py --config c4 --steps 20 --warmup 5 > $O/bench_c4.json 2> $O/bench_c4.err && echo bench_c4 ok
python bench.py --force-dist --steps 21 --warmup 5 --no-cpu-baseline --no-side-legs > $O/bench_fd.json 2> $O/bench_fd.err && echo bench_fd ok
GN_PRECISION=bf16x6 python bench.py --no-cpu-baseline --no-side-legs > $O/bench_c2_bf16x6.json 2> $O/bench_c2_bf16x6.err && echo bench_x6 ok
for CFG in c2 c4; do
  C="--config $CFG --steps 20 --warmup 5 --no-cpu-baseline --no-side-legs"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${CFG}_s1 -- python3 bench.py $C --streams 1 > $O/${CFG}_s1_line.json 2> $O/${CFG}_s1.err && echo ${CFG}_s1 ok
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${CFG}_fetch -- python3 bench.py $C --streams 1 > /dev/null 2> $O/${CFG}_fetch.err && echo ${CFG}_fetch ok
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${CFG}_write -- python3 bench.py $C --streams 1 > /dev/null 2> $O/${CFG}_write.err && echo ${CFG}_write ok
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c2_dflt -- python3 bench.py --config c2 --steps 20 --warmup 5 --no-cpu-baseline --no-side-legs > $O/c2_dflt_line.json 2> $O/c2_dflt.err && echo c2_dflt ok
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c2_train -- python3 tools/train_step_time.py > $O/c2_train.log 2>&1 && echo train ok
bash tools/profile_sq.sh c2 > $O/sq_c2.txt 2>&1; bash tools/profile_sq.sh c4 > $O/sq_c4.txt 2>&1; echo sq ok
python tools/diag/stamps_fwd.py stamps.bin 512 11 > $O/stamps_c2.txt 2>&1; echo stamps ok
du -sh $O
