#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03_direct3; rm -rf $OUT; mkdir -p $OUT
L=conv1.3,dec1.0,dec2.2
ONLY=$L CHECK=1 SUSTAIN=20 SEGK_NO_DIRECT=1 timeout -k 10 200 python3 scratch/conv_bench.py fwd dgrad > $OUT/old.txt 2>&1 || { tail $OUT/old.txt; exit 1; }
ONLY=$L CHECK=1 SUSTAIN=20 SEGK_DIRECT_NO_PAIRS=1 timeout -k 10 200 python3 scratch/conv_bench.py fwd dgrad > $OUT/nopairs.txt 2>&1 || { tail $OUT/nopairs.txt; exit 1; }
ONLY=$L CHECK=1 SUSTAIN=20 timeout -k 10 200 python3 scratch/conv_bench.py fwd dgrad > $OUT/new.txt 2>&1 || { tail $OUT/new.txt; exit 1; }
grep -h "\^2" $OUT/old.txt $OUT/nopairs.txt $OUT/new.txt | cut -c1-300
timeout -k 10 900 python3 -u -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py -x -q --timeout 300 --timeout-method thread > $OUT/pytest.txt 2>&1 || { tail -40 $OUT/pytest.txt; exit 1; }
tail -2 $OUT/pytest.txt
for i in 1 2; do
  SEGK_DIRECT_NO_PAIRS=1 timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --launch-table $OUT/launch_old.json >> $OUT/bench_nopairs.json 2>> $OUT/err.txt || exit 1
  timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --launch-table $OUT/launch_new.json >> $OUT/bench_new.json 2>> $OUT/err.txt || exit 1
done
python3 - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03_direct3/bench_*.json')):
    for line in open(f):
        d=json.loads(line); r=d['roofline']; print(f.split('/')[-1], d['value'], d['ms_per_step'], r['conv_ms_per_step'], r['by_kernel'].get('conv_fwd_kernel[narrow]'))
a=json.load(open('gpurun_out/r03_direct3/launch_old.json')); b=json.load(open('gpurun_out/r03_direct3/launch_new.json'))
for x,y in zip(a,b):
    if abs(x['us_mean']-y['us_mean'])>8: print(x['i'],x['kind'],x['shape'],x['us_mean'],'->',y['us_mean'])
P
