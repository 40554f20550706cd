export TMPDIR=/tmp
mkdir -p gpurun_out/r4_split
O=gpurun_out/r4_split
timeout 900 python -m pytest tests/test_gpu_sa_fused.py -q -m gpu -x 2>&1 | tail -3
timeout 600 python tools/sa_bench.py 2>&1 | tail -6 | tee $O/sa_bench_v4.txt
timeout 600 python tools/secondary_bench.py 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
for k,v in d.items():
    if isinstance(v,dict): print(k, v.get('value'), v.get('ms'), (v.get('roofline') or {}).get('achieved'), v.get('error'))
" | tee $O/secondary_v4.txt
