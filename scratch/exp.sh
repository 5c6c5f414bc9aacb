#!/bin/bash
# usage: exp.sh VARIANT  -> runs conv_bench fwd with libsegk_VARIANT.so swapped in
cd $GRAFT_REPO_ROOT
cp image-segmentation_amd/libsegk.so /tmp/libsegk_orig.so
for v in "$@"; do
  cp image-segmentation_amd/libsegk_$v.so image-segmentation_amd/libsegk.so
  echo "== $v"; timeout -k 10 200 python scratch/conv_bench.py fwd 2>&1 | grep -E "conv3.17|conv4.27|dec3.0|conv1.3|dec1.0|TOTAL"
done
cp /tmp/libsegk_orig.so image-segmentation_amd/libsegk.so
