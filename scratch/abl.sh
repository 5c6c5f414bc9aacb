#!/bin/bash
cd $GRAFT_REPO_ROOT
for d in ${DBGS:-0 7 15}; do
  echo "== SEGK_DBG=$d"
  SEGK_DBG=$d ONLY=conv1.3,dec1.0 timeout -k 10 120 python scratch/conv_bench.py fwd 2>&1 | grep -E "conv1.3|dec1.0" || exit 1
done
