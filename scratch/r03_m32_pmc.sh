cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export ONLY=conv3.17
for v in m32 m16; do
O=gpurun_out/pmc_$v; rm -rf $O; mkdir -p $O
if [ $v = m16 ]; then export SEGK_NO_M32=1; fi
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $O -o p1 -- python3 scratch/conv_bench.py fwd > $O/p1.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA --output-format csv -d $O -o p2 -- python3 scratch/conv_bench.py fwd > $O/p2.log 2>&1 || exit 1
find $O -name '*counter_collection.csv' | xargs python3 scratch/pmcsum.py > $O/summary.txt
echo == $v; cat $O/summary.txt
done
