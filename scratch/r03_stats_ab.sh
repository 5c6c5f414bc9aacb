set -e
mkdir -p gpurun_out/r03_stats
O=gpurun_out/r03_stats
export ONLY=conv1.3,conv2.7 CHECK=1 SUSTAIN=300
timeout -k 10 120 python scratch/conv_bench.py fwds > $O/old_natural.txt 2>&1
FINE=1 SEGK_NO_DIRECT=1 timeout -k 10 120 python scratch/conv_bench.py fwds > $O/old_fine.txt 2>&1
FINE=1 timeout -k 10 120 python scratch/conv_bench.py fwds > $O/direct_fine.txt 2>&1
timeout -k 10 120 python scratch/conv_bench.py fwd > $O/direct_nostats.txt 2>&1
cat $O/old_natural.txt $O/old_fine.txt $O/direct_fine.txt $O/direct_nostats.txt
