set -e
O=gpurun_out/r03_mask; mkdir -p $O
export ONLY=dec3.2,dec4.2,dec5.2,center.2,center.0,dec2.0,dec3.0,dec4.0,dec5.0 CHECK=1 SUSTAIN=100 BIAS=1 MASK=1
for round in 1 2; do
SEGK_LIB_PATH=$PWD/image-segmentation_amd/libsegk_exp.so timeout -k 10 120 python scratch/conv_bench.py fwd dgrad > $O/base_$round.txt 2>&1
timeout -k 10 120 python scratch/conv_bench.py fwd dgrad > $O/prefetch_$round.txt 2>&1
done
cat $O/base_1.txt $O/prefetch_1.txt $O/base_2.txt $O/prefetch_2.txt | grep -v amdgpu.ids | cut -c1-200
