export TMPDIR=/tmp
mkdir -p gpurun_out/r4_split
O=gpurun_out/r4_split
timeout 900 python -m pytest tests/test_gpu_sa_fused.py -q -m gpu -x 2>&1 | tail -5
for t in 128 0; do echo "== TGN_SA_TILE=$t"; TGN_SA_TILE=$t timeout 600 python tools/sa_bench.py 2>&1 | tail -6; done | tee $O/sa_bench_v3.txt
echo "== bench --shape B --fused 1"; timeout 600 python bench.py --shape B --fused 1 --steps 5 --warmup 2 --cpu-meshes 0 --no-alt 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['kernel_ms_per_step'])" | tee $O/bench_fused_B_v3.txt
echo "== bench --shape A --fused 1"; timeout 600 python bench.py --shape A --fused 1 --steps 5 --warmup 2 --cpu-meshes 0 --no-alt 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['kernel_ms_per_step'])" | tee $O/bench_fused_A_v3.txt
