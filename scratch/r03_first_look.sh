#!/bin/bash
# In-step per-launch trace, event table, sustained layer table and the streaming ceiling probe, one box.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03_first; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 240 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-wgrad-stream --launch-table $OUT/launch_table.json > $OUT/bench_one_stream.json 2> $OUT/bench.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o trace -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-wgrad-stream > $OUT/bench_under_rocprof.json 2> $OUT/trace.err || exit 1
SUSTAIN=40 timeout -k 10 300 python3 scratch/conv_bench.py > $OUT/sustained.txt 2> $OUT/sustained.err || exit 1
timeout -k 10 200 scratch/stream_ceiling > $OUT/stream_ceiling.txt 2>&1 || exit 1
find $OUT -name '*kernel_trace.csv' | head -1 | xargs -I{} python3 scratch/instep_table.py {} $OUT/launch_table.json $OUT/sustained.txt > $OUT/instep_vs_sustained.txt 2> $OUT/instep.err
ls -la $OUT; tail -5 $OUT/instep_vs_sustained.txt
