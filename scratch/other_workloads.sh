#!/bin/bash
# bench lines of the other workloads on one box -> gpurun_out/other_workloads/<name>.json
OUT=gpurun_out/other_workloads; rm -rf $OUT; mkdir -p $OUT
run() { name=$1; shift; timeout -k 10 240 python3 bench.py --no-cpu-baseline --steps 12 --warmup 4 "$@" > $OUT/$name.json 2> $OUT/$name.err || { echo "$name failed"; tail -3 $OUT/$name.err; exit 1; }; echo "$name: $(cut -c1-160 $OUT/$name.json)"; }
run unetpp_batch_16 --model unetpp && run unetpp_deepsup_batch_16 --model unetpp --deep-sup && run unetvgg_convtrans --model unetvgg_convtrans && \
run unetsimple_batch_16 --model unetsimple && run spidernet_batch_8 --model spidernet && run detectmid_batch_12 --model detectmid
