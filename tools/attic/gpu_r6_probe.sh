#!/bin/bash
# round 6: FPS levels 2-3 on the lean register-resident kernel ("fps_lean") against the bucket / plain ones, in the step and alone
export TMPDIR=/tmp
O=gpurun_out/r6_lean
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_modules.py tests/test_gpu_hotpath_configs.py tests/test_gpu_fps_prefix.py tests/test_gpu_random_sweep.py -x -q -k "fps or furthest or hot or Hot or sample_and_group or shape or prefix or certificate" 2>&1 | tail -4
run() {  # name, lean, args...
  local name=$1 lean=$2; shift; shift
  TGN_FPS_LEAN=$lean timeout 600 python bench.py --steps 20 --warmup 5 --cpu-meshes 0 --secondary 0 --no-alt "$@" 2>$O/$name.err | tail -1 > $O/$name.json
  python -c "import json; d=json.load(open('$O/$name.json')); print('$name', round(d['value']), round(d['ms_per_step'],4), d['kernel_ms_per_step'])" 2>&1 | tail -1
}
run lean0_bucket 0
run lean1_bucket 1
run lean2_plain 2 --fps23 plain
run lean1_bucket_b 1
run lean2_plain_b 2 --fps23 plain
for L in 0 1; do echo "phase2 alone, bucket l2, lean=$L"; TGN_FPS_LEAN=$L timeout 600 python tools/phase2_bench.py 2>/dev/null | tail -1; done
for L in 2; do echo "phase2 alone, plain, lean=$L"; TGN_FPS_LEAN=$L TGN_FPS_BUCKET_MIN=4097 timeout 600 python tools/phase2_bench.py 2>/dev/null | tail -1; done
