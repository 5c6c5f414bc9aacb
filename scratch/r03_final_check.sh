cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_final; rm -rf $O; mkdir -p $O
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1 || { tail -20 $O/smoke.txt; exit 1; }
tail -2 $O/smoke.txt
timeout -k 10 400 python3 bench.py > $O/bench.json 2> $O/bench.err || { tail $O/bench.err; exit 1; }
python3 -c "
import json; d=json.load(open('$OUT/bench.json' if False else 'gpurun_out/r03_final/bench.json')); r=d['roofline']; print(d['value'], d['ms_per_step'], r['frac'], r['traffic'], r['traffic_note'][:80], d['cpu_baseline']['value'])"
