#!/bin/bash
# HBM traffic of the gather family (tools/gather_family_variants.py 5): FETCH_SIZE and WRITE_SIZE in separate counter passes, kernel-trace only.
set -u
export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/g_$C -o pmc -- python $GRAFT_REPO_ROOT/tools/gather_family_variants.py 5 > /dev/null 2>&1)
done
python - <<'PY'
import csv, glob, collections
res = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"/tmp/g_{c}/*counter_collection.csv")[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == c:
            k = r["Kernel_Name"].split("(")[0].replace("void tgn::", "").replace("tgn::", "")
            if any(t in k for t in ("gather_rows", "grouping", "subtraction", "aggregation", "interpolation")):
                agg[k].append(float(r["Counter_Value"]) * 1024 / 1e6)
    for k, v in agg.items():
        v = sorted(v)
        res[k][c] = v[len(v) // 2]
print("# HBM traffic per launch of the gather family at (n, nsample, c, w_c) = (24000, 36, 32, 4), MB: median over the launches of")
print("# tools/gather_family_variants.py 5; rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, kernel-trace only, counter x 1024")
for k, v in sorted(res.items()):
    print("%-52s fetch %8.1f   write %8.1f" % (k, v.get("FETCH_SIZE", 0), v.get("WRITE_SIZE", 0)))
PY
