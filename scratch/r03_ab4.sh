#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03_ab4; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 600 python3 -u -m pytest tests/test_gpu_ops.py tests/test_gpu_volume.py -x -q --timeout 120 --timeout-method thread > $OUT/pytest.txt 2>&1 || { tail -30 $OUT/pytest.txt; exit 1; }
tail -2 $OUT/pytest.txt
for i in 1 2; do
  SEGK_WGRAD_REDUCE_SCALAR=1 timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline >> $OUT/bench_scalar.json 2>> $OUT/err.txt || exit 1
  timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline >> $OUT/bench_vec.json 2>> $OUT/err.txt || exit 1
done
python3 - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03_ab4/bench_*.json')):
    for line in open(f):
        d=json.loads(line); print(f.split('/')[-1], d['value'], d['ms_per_step'], d['roofline']['conv_ms_per_step'], d['roofline']['wgrad_ms'])
P
