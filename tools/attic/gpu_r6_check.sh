#!/bin/bash
# round 6: full GPU suite, the driver's bench command, the FPS floors, the direct-form SA kernel in both arithmetic forms
set -u
export TMPDIR=/tmp
O=gpurun_out/r6_check
mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
tools/_bin/fps_floor > $O/fps_floor.txt 2>&1; cat $O/fps_floor.txt | cut -c1-400
timeout 300 python tools/sa_direct_time.py 2>&1 | tail -6 | tee $O/sa_direct.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_20.json 2> $O/bench_20.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r6_check/bench_20.json"))
print(round(d["value"]), round(d["ms_per_step"],4), d["kernel_ms_per_step"], d.get("kernel_timing"))
r=d["roofline"]; print({k:r.get(k) for k in ("achieved","peak","frac","primitive_floor_us","frac_vs_primitive_floor","frac_vs_chain_plus_one_bucket","traffic")})
s=d.get("secondary",{})
for k in ("shape_A_tree_ties","shape_A_with_h2d"):
    print(k, json.dumps(s.get(k))[:700])
print({k:(v.get("value"),v.get("ms"),v.get("error")) for k,v in s.items() if isinstance(v,dict)})
print(d["config"]["schedule"])
PY
