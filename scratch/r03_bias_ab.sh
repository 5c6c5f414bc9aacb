set -e
O=gpurun_out/r03_bias; mkdir -p $O
export ONLY=conv1.3,dec1.0 CHECK=1 SUSTAIN=300 BIAS=1 FINE=1
for round in 1 2; do
SEGK_LIB_PATH=$PWD/image-segmentation_amd/libsegk_exp.so timeout -k 10 120 python scratch/conv_bench.py fwds fwd > $O/global_bias_$round.txt 2>&1
timeout -k 10 120 python scratch/conv_bench.py fwds fwd > $O/lds_bias_$round.txt 2>&1
done
BIAS= timeout -k 10 120 python scratch/conv_bench.py fwds fwd > $O/no_bias.txt 2>&1
cat $O/global_bias_1.txt $O/lds_bias_1.txt $O/global_bias_2.txt $O/lds_bias_2.txt $O/no_bias.txt | grep -v amdgpu.ids | cut -c1-120
