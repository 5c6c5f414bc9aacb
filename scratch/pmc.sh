#!/bin/bash
# usage: pmc.sh TAG MODE LAYERS   -- three PMC passes over conv_bench.py (one pass per run)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=$1; export ONLY=$3
mkdir -p gpurun_out/pmc_$TAG
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_$TAG -o p1 -- python3 scratch/conv_bench.py $2 > gpurun_out/pmc_$TAG/p1.log 2>&1 &&
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM --output-format csv -d gpurun_out/pmc_$TAG -o p2 -- python3 scratch/conv_bench.py $2 > gpurun_out/pmc_$TAG/p2.log 2>&1
ls gpurun_out/pmc_$TAG
