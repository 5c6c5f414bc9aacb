#!/bin/bash
# same-box A/B: (1) weight-gradient wave layout 2x2 vs 1x4, (2) non-temporal accesses in the normalisation passes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03_ab1; rm -rf $OUT; mkdir -p $OUT
EXP=$GRAFT_REPO_ROOT/image-segmentation_amd/libsegk_exp.so
for L in 22 14 22 14; do
  SEGK_WGRAD_LAYOUT=$L CHECK=1 SUSTAIN=30 timeout -k 10 200 python3 scratch/conv_bench.py wgrad >> $OUT/wgrad_layout_$L.txt 2>&1 || exit 1
done
timeout -k 10 100 python3 scratch/stream_bench.py norm > $OUT/norm_plain.txt 2>&1 || exit 1
SEGK_LIB_PATH=$EXP timeout -k 10 100 python3 scratch/stream_bench.py norm > $OUT/norm_nt.txt 2>&1 || exit 1
for i in 1 2; do
  timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline >> $OUT/bench_plain.json 2>> $OUT/err.txt || exit 1
  SEGK_LIB_PATH=$EXP timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline >> $OUT/bench_nt.json 2>> $OUT/err.txt || exit 1
  SEGK_WGRAD_LAYOUT=14 timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline >> $OUT/bench_wg14.json 2>> $OUT/err.txt || exit 1
done
grep -h TOTAL $OUT/wgrad_layout_*.txt; cat $OUT/norm_plain.txt $OUT/norm_nt.txt
python3 - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03_ab1/bench_*.json')):
    for line in open(f):
        d=json.loads(line); print(f.split('/')[-1], d['value'], d['ms_per_step'], d['roofline']['conv_ms_per_step'], d['roofline']['wgrad_ms'])
P
