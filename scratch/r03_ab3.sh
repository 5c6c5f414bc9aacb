#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03_ab3; rm -rf $OUT; mkdir -p $OUT
EXP=$GRAFT_REPO_ROOT/image-segmentation_amd/libsegk_exp.so
SEGK_STREAM_BIG=0 timeout -k 10 100 python3 scratch/stream_bench.py norm 2>&1 | grep "^norm" > $OUT/norm_plain.txt || exit 1
timeout -k 10 100 python3 scratch/stream_bench.py norm 2>&1 | grep "^norm" > $OUT/norm_u1.txt || exit 1
SEGK_LIB_PATH=$EXP timeout -k 10 100 python3 scratch/stream_bench.py norm 2>&1 | grep "^norm" > $OUT/norm_u2.txt || exit 1
for i in 1 2; do
  SEGK_STREAM_BIG=0 timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline >> $OUT/bench_plain.json 2>> $OUT/err.txt || exit 1
  timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline >> $OUT/bench_u1.json 2>> $OUT/err.txt || exit 1
  SEGK_LIB_PATH=$EXP timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline >> $OUT/bench_u2.json 2>> $OUT/err.txt || exit 1
done
tail -n 4 $OUT/norm_plain.txt $OUT/norm_u1.txt $OUT/norm_u2.txt
python3 - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03_ab3/bench_*.json')):
    for line in open(f):
        d=json.loads(line); print(f.split('/')[-1], d['value'], d['ms_per_step'], d['roofline']['conv_ms_per_step'], d['roofline']['wgrad_ms'])
P
