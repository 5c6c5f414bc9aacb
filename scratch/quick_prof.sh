#!/bin/bash
# Kernel trace of a short default bench run (per-kernel times only).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_quick; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o trace -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/trace.err || exit 1
