# same-box A/B of two builds of the library: bench.py with libsegk_exp.so (A) and libsegk.so (B), alternating
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_ablib; rm -rf $O; mkdir -p $O
for r in 1 2; do
SEGK_LIB_PATH=$PWD/image-segmentation_amd/libsegk_exp.so timeout -k 10 200 python3 bench.py --no-cpu-baseline > $O/A_$r.json 2> $O/A_$r.err || exit 1
timeout -k 10 200 python3 bench.py --no-cpu-baseline > $O/B_$r.json 2> $O/B_$r.err || exit 1
done
python3 - <<'PY'
import json
for n in ("A_1","B_1","A_2","B_2"):
    d=json.load(open(f"gpurun_out/r03_ablib/{n}.json")); r=d["roofline"]
    print(n, d["value"], d["ms_per_step"], r["frac"], r["conv_ms_per_step"], {k:v.get("frac") for k,v in r.get("by_kind",{}).items()} if isinstance(r.get("by_kind"),dict) else "")
PY
