#!/bin/bash
# big-tensor access pattern in the normalisation passes: parity with it forced on, micro-benchmark, in-step A/B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03_ab2; rm -rf $OUT; mkdir -p $OUT
SEGK_STREAM_BIG=1 timeout -k 10 600 python3 -u -m pytest tests/test_gpu_ops.py tests/test_zz_no_alloc.py -x -q --timeout 120 --timeout-method thread > $OUT/pytest_big.txt 2>&1 || { tail -30 $OUT/pytest_big.txt; exit 1; }
tail -3 $OUT/pytest_big.txt
SEGK_STREAM_BIG=0 timeout -k 10 100 python3 scratch/stream_bench.py norm 2>&1 | grep "^norm" > $OUT/norm_plain.txt || exit 1
timeout -k 10 100 python3 scratch/stream_bench.py norm 2>&1 | grep "^norm" > $OUT/norm_auto.txt || exit 1
for i in 1 2; do
  SEGK_STREAM_BIG=0 timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline >> $OUT/bench_plain.json 2>> $OUT/err.txt || exit 1
  timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline >> $OUT/bench_auto.json 2>> $OUT/err.txt || exit 1
done
cat $OUT/norm_plain.txt $OUT/norm_auto.txt
python3 - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03_ab2/bench_*.json')):
    for line in open(f):
        d=json.loads(line); print(f.split('/')[-1], d['value'], d['ms_per_step'], d['roofline']['conv_ms_per_step'], d['roofline']['wgrad_ms'])
P
