#!/bin/bash
# the whole GPU suite, then the default bench
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03_tb; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 900 python3 -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > $OUT/pytest_gpu.txt 2>&1 || { tail -40 $OUT/pytest_gpu.txt; exit 1; }
tail -3 $OUT/pytest_gpu.txt
timeout -k 10 300 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err || { tail $OUT/bench.err; exit 1; }
cat $OUT/bench.json
