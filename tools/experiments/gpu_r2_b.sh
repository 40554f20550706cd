#!/bin/bash
# round 2, call B: debug the v2 grouping kernel against v1, L2/EA counters of the grouping variants
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
python - <<'PY'
import sys, numpy as np, torch
sys.path.insert(0, '.')
from toothgroupnetwork_amd import _lib
L = _lib.lib(); dev = torch.device('cuda')
def run(B,N,S,K,D,xf,impl,idt):
    g = torch.Generator().manual_seed(1)
    xyz = torch.randn(B,N,3,generator=g).to(dev); pts = torch.randn(B,N,max(D,1),generator=g)[:,:,:D].contiguous().to(dev)
    new_xyz = xyz[:,:S].contiguous(); idx = torch.randint(0,N,(B,S,K),generator=g).to(idt).to(dev)
    out = torch.full((B,S,K,3+D), float('nan'), device=dev)
    _lib.check(L.tgn_group_points_ex(B,N,S,K,D,_lib.ptr(xyz),_lib.ptr(new_xyz),_lib.ptr(pts) if D else None,_lib.ptr(idx),int(idt==torch.int64),int(xf),_lib.ptr(out),impl,16,0,_lib.stream()))
    torch.cuda.synchronize()
    return out
for (B,N,S,K,D) in [(1,50,1,4,1),(1,50,2,8,6),(2,300,37,32,6),(2,4096,512,32,6),(2,300,37,32,128),(2,300,37,64,253),(9,1024,256,32,512)]:
    for xf in (1,0):
        for idt in (torch.int32, torch.int64):
            a = run(B,N,S,K,D,xf,1,idt); b = run(B,N,S,K,D,xf,2,idt)
            bad = ~((a==b) | (a.isnan() & b.isnan()))
            nb = int(bad.sum())
            msg = ''
            if nb:
                w = bad.nonzero()[:6].tolist()
                msg = f' first bad (b,s,k,c): {w} got {[float(b[tuple(i)]) for i in w[:3]]} want {[float(a[tuple(i)]) for i in w[:3]]}; bad per c: {bad.sum((0,1,2)).tolist()[:16]} nan in v2: {int(b.isnan().sum())}'
            print((B,N,S,K,D), 'xyz_first', xf, str(idt)[6:], 'mismatches', nb, '/', a.numel(), msg, flush=True)
PY
i=0
for SET in "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum" "TCC_READ_sum TCC_WRITE_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_64B_sum"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmcg2_$i -o pmc -- \
      python $GRAFT_REPO_ROOT/tools/group_bench.py quick > $GRAFT_REPO_ROOT/gpurun_out/pmcg2_$i.log 2>&1)
  tail -3 gpurun_out/pmcg2_$i.log
done
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("gpurun_out/pmcg2_*/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        if "group_points" not in k: continue
        agg[(k, row["Grid_Size"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
for (k, g), cs in sorted(agg.items()):
    print(k, "grid", g)
    for c, v in sorted(cs.items()):
        print(f"   {c:32s} n={len(v):2d} mean={sum(v)/len(v):16.1f}")
PY
