cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_m32; rm -rf $O; mkdir -p $O
timeout -k 10 600 python3 -u -m pytest tests/test_gpu_ops.py -x -q --timeout 120 --timeout-method thread > $O/pytest_ops.txt 2>&1 || { tail -40 $O/pytest_ops.txt; exit 1; }
tail -2 $O/pytest_ops.txt
export CHECK=1 SUSTAIN=100 BIAS=1
SEGK_LIB_PATH=$PWD/image-segmentation_amd/libsegk_exp.so timeout -k 10 200 python3 scratch/conv_bench.py fwd dgrad > $O/base.txt 2>&1 || exit 1
timeout -k 10 200 python3 scratch/conv_bench.py fwd dgrad > $O/m32.txt 2>&1 || exit 1
SEGK_NO_M32=1 timeout -k 10 200 python3 scratch/conv_bench.py fwd dgrad > $O/m16_newswz.txt 2>&1 || exit 1
paste -d'\n' $O/base.txt $O/m32.txt $O/m16_newswz.txt | grep -v amdgpu | cut -c1-150
