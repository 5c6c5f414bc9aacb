#!/bin/bash
# the whole GPU suite, then the default bench, then the one-rank forced-collective bench
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03_tb; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 1000 python3 -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $OUT/pytest_gpu.txt 2>&1 || { tail -60 $OUT/pytest_gpu.txt; exit 1; }
tail -3 $OUT/pytest_gpu.txt
timeout -k 10 300 python3 bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err || { tail $OUT/bench.err; exit 1; }
python3 -c "
import json; d=json.load(open('$OUT/bench.json')); r=d['roofline']; print(d['value'], d['ms_per_step'], r['frac'], r['conv_ms_per_step'], r['conv_ms_per_step_raw_events'], r['traffic_note'])"
SEGK_FORCE_COLLECTIVES=1 timeout -k 10 300 python3 bench.py --no-cpu-baseline > $OUT/bench_forced.json 2> $OUT/bench_forced.err || { tail $OUT/bench_forced.err; exit 1; }
python3 -c "
import json; d=json.load(open('$OUT/bench_forced.json')); print(d['value'], d['ms_per_step'], json.dumps(d['config']['collective']))"
