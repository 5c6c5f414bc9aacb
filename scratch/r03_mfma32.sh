set -e
O=gpurun_out/r03_mfma32; mkdir -p $O
for w in 1 2; do
timeout -k 10 60 scratch/mfma_ceiling $w 10000 16 >> $O/out.txt
timeout -k 10 60 scratch/mfma_ceiling $w 10000 32 >> $O/out.txt
done
cat $O/out.txt
