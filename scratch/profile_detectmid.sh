#!/bin/bash
# Kernel trace of the 3-D classifier workload (bench.py --model detectmid).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_dm; mkdir -p $OUT
timeout -k 10 400 python3 bench.py --model detectmid --steps 10 --warmup 3 > $OUT/bench_n1.json 2> $OUT/bench_n1.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o trace -- python3 bench.py --model detectmid --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/trace.err || exit 1
ls $OUT
