#!/bin/bash
# Dev tool (GPU box): small-M Linear -- stream kernel tests + engine tests, then bench.py's small-M table only
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r5n1}
mkdir -p $O
: > $O/tests.txt
tail -3 $O/tests.txt
:
tail -3 $O/tests.txt
timeout 600 python bench.py --layers 2 --steps 2 --warmup 1 --no-cpu-baseline --no-config5 --no-config4 --no-extra-modes --no-dropin > $O/b.json 2> $O/b.err
python - <<PY
import json
for ln in open("$O/b.json"):
    if ln.startswith("{"): d = json.loads(ln)
sm = d["roofline_linear_eps_smallm"]
print("headline", round(sm["frac"], 3), sm["kernel"][-40:])
for k in ("table_gate_up_sized", "table_down_sized"):
    print(k)
    for t in sm[k]:
        print("  M %4d fwd %6.1f us | dgrad %6.1f us | pair %.3f" % (t["M"], t["fwd_us"], t["dgrad_us"], t["pair_frac"]))
PY
