set -u
timeout 600 python -m pytest tests/test_gpu_pt_attention.py tests/test_gpu_whole_nets.py -m gpu -q -x 2>&1 | tail -3
timeout 300 python tools/pt_forward_bench.py 2>&1 | grep -v amdgpu.ids | head -3
python tools/pt_profile.py 2>&1 | grep -v amdgpu.ids | cut -c1-75,150-215 | grep "softmax_aggregate\|Self CUDA"
python tools/train_step_bench.py --graph 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:(round(v['ms_per_step'],2) if isinstance(v,dict) else v) for k,v in d.items() if k!='workload'})"
