#!/bin/bash
# same-box A/B of one environment switch: bash scratch/ab_env.sh VAR=VALUE   (prints slices/s, ms/step, conv ms/step)
run() { echo -n "[$1] "; env $1 timeout -k 10 120 python bench.py --no-cpu-baseline --steps 32 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['conv_ms_per_step'], d['config']['final_loss'])"; }
run "SEGK_AB=base" && run "$1" && run "SEGK_AB=base" && run "$1"
