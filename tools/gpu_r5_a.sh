#!/bin/bash
# round 5, first GPU call: the whole -m gpu suite, the FPS latency floor, the driver's bench command (with `secondary`, incl. the
# gather family), the per-kernel table of the PointNet++ forward
set -u
export TMPDIR=/tmp
O=gpurun_out/r5a
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
timeout 300 python tools/fps_floor_report.py > $O/fps_floor.txt 2> $O/fps_floor.err; echo "floor rc=$?"; cat $O/fps_floor.txt | cut -c1-220; tail -3 $O/fps_floor.err
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_20.json 2> $O/bench_20.err; echo "bench rc=$?"; tail -2 $O/bench_20.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5a/bench_20.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "n_gpus", "backend", "rccl_version")})
print("roofline", {k: d["roofline"].get(k) for k in ("achieved", "peak", "frac", "frac_vs_chain_plus_one_bucket")}, d["roofline"].get("floor"))
print("kernel_ms", d.get("kernel_ms_per_step"))
s = d.get("secondary", {})
for k, v in s.items():
    if k == "gather_family":
        for kk, vv in v.items():
            if isinstance(vv, dict):
                print("  ", kk, round(vv["us"], 2), "us", round(vv["roofline"]["achieved"]), "GB/s", round(vv["roofline"]["frac"], 3))
    elif isinstance(v, dict):
        print(k, {a: v[a] for a in ("value", "ms", "error", "skipped") if a in v}, (v.get("roofline") or {}).get("frac"))
    else:
        print(k, v)
PY
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/pnpp -o pnpp -- python $GRAFT_REPO_ROOT/tools/pnpp_forward_run.py > $GRAFT_REPO_ROOT/$O/pnpp_run.log 2>&1)
tail -1 $O/pnpp_run.log
f=$(find $O/pnpp -name "*kernel_stats.csv" | head -1); cp $f $O/pnpp_forward_kernel_stats.csv; cut -c1-160 $O/pnpp_forward_kernel_stats.csv | head -45
rm -rf $O/pnpp
