#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03_parity; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 600 python3 -u -m pytest tests/test_gpu_models.py -m gpu -x -q -s --timeout 300 --timeout-method thread -k "well_conditioned or bench_shape" > $OUT/pytest.txt 2>&1 || { tail -40 $OUT/pytest.txt; exit 1; }
grep -E "wellcond|passed|failed" $OUT/pytest.txt | cut -c1-400
timeout -k 10 600 python3 scratch/wellcond_bf16.py > $OUT/wellcond_bf16.txt 2>&1 || { tail -20 $OUT/wellcond_bf16.txt; exit 1; }
cat $OUT/wellcond_bf16.txt
