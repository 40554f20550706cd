export TMPDIR=/tmp
mkdir -p gpurun_out/r4_split
O=gpurun_out/r4_split
timeout 900 python -m pytest tests/test_gpu_sa_fused.py -q -s -m gpu -k "two_layer_level_vs_oracle" 2>&1 | grep -E "bf16x3|fp32-mfma|passed|failed|Error|error" | cut -c1-300 > $O/sa_fused_errors.txt
tail -30 $O/sa_fused_errors.txt
for v in 1 0; do echo "== TGN_SA_BF16X3=$v"; TGN_SA_BF16X3=$v timeout 600 python tools/sa_bench.py 2>&1 | tail -12; done | tee $O/sa_bench.txt
timeout 600 python -m pytest tests/test_gpu_sa_fused.py tests/test_gpu_whole_nets.py tests/test_gpu_r4_parity.py tests/test_gpu_modules.py -q -m gpu 2>&1 | tail -8
for v in 1 0; do echo "== bench --shape B --fused 1, TGN_SA_BF16X3=$v"; TGN_SA_BF16X3=$v timeout 600 python bench.py --shape B --fused 1 --steps 5 --warmup 2 --cpu-meshes 0 --no-alt 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['kernel_ms_per_step'])"; done | tee $O/bench_fused_B.txt
