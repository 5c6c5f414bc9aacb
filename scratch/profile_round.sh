#!/bin/bash
# (the profiler legs keep every kernel on one stream: overlapping kernels stretch each other's durations)
# Kernel-trace + PMC passes (HBM read bytes, HBM write bytes, MFMA busy cycles) of the default bench; one program run per pass.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_round; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > $OUT/bench_n1.json 2> $OUT/bench_n1.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o trace -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-wgrad-stream > $OUT/bench_under_rocprof.json 2> $OUT/trace.err || exit 1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT -o fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-wgrad-stream > $OUT/fetch.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT -o write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-wgrad-stream > $OUT/write.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT -o mfma -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-wgrad-stream > $OUT/mfma.log 2>&1 || exit 1
ls -la $OUT
