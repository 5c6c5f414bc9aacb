#!/bin/bash
# Kernel trace of a short bench run of another workload: quick_prof_model.sh MODEL [extra bench args]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
M=$1; shift
OUT=gpurun_out/prof_$M; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o trace -- python3 bench.py --model $M --steps 5 --warmup 2 --no-cpu-baseline "$@" > $OUT/bench_under_rocprof.json 2> $OUT/trace.err || exit 1
