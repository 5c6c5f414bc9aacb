#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03_direct; rm -rf $OUT; mkdir -p $OUT
L=conv1.3,dec1.0,dec2.2,conv2.7
ONLY=$L CHECK=1 SUSTAIN=20 SEGK_NO_DIRECT=1 timeout -k 10 200 python3 scratch/conv_bench.py fwd dgrad > $OUT/old.txt 2>&1 || { tail $OUT/old.txt; exit 1; }
ONLY=$L CHECK=1 SUSTAIN=20 timeout -k 10 200 python3 scratch/conv_bench.py fwd dgrad > $OUT/new.txt 2>&1 || { tail $OUT/new.txt; exit 1; }
grep -h "\^2" $OUT/old.txt $OUT/new.txt | cut -c1-400
