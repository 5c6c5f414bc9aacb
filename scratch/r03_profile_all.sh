#!/bin/bash
# Round profile on ONE box: default bench (with CPU baseline), kernel trace, three PMC passes, in-step table inputs, the
# other workloads and the 3-D classifier trace.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash scratch/profile_round.sh || exit 1
OUT=gpurun_out/prof_round
timeout -k 10 240 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-wgrad-stream --launch-table $OUT/launch_table.json > $OUT/bench_one_stream.json 2> $OUT/bench_one_stream.err || exit 1
SUSTAIN=40 BIAS=1 timeout -k 10 300 python3 scratch/conv_bench.py > $OUT/sustained.txt 2> $OUT/sustained.err || exit 1
SEGK_FORCE_COLLECTIVES=1 timeout -k 10 300 python3 bench.py --no-cpu-baseline > $OUT/bench_forced_one_rank.json 2> $OUT/bench_forced.err || exit 1
bash scratch/other_workloads.sh || exit 1
bash scratch/profile_detectmid.sh || exit 1
ls $OUT gpurun_out/other_workloads gpurun_out/prof_dm
