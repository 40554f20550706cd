export TMPDIR=/tmp
mkdir -p gpurun_out/r4_ball
O=gpurun_out/r4_ball
echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null) ; affinity $(python -c 'import os; print(len(os.sched_getaffinity(0)))') ; effective $(python -c 'from toothgroupnetwork_amd import sharding; print(sharding.effective_cpus())')" | tee $O/cpu.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_modules.py tests/test_gpu_random_sweep.py tests/test_gpu_hotpath_configs.py -q -m gpu -x -k "ball or full_size or hotpath or sweep or pipelined or shape_b" 2>&1 | tail -5
for pair in 1 0; do
  echo "== TGN_BALL_PAIR=$pair"
  TGN_BALL_PAIR=$pair timeout 600 python bench.py --steps 20 --warmup 5 --cpu-meshes 0 --no-alt --secondary 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['kernel_ms_per_step'])"
  TGN_BALL_PAIR=$pair timeout 600 python bench.py --steps 20 --warmup 5 --cpu-meshes 0 --no-alt --secondary 0 --pipeline 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('one stream:', d['value'], d['ms_per_step'], d['kernel_ms_per_step'])"
done | tee $O/bench_pair.txt
