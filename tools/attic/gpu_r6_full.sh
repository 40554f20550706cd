#!/bin/bash
# round 6: the driver's own sequence -- GPU tests, smoke, the default bench line (with secondary) -- and a 20-step bench
export TMPDIR=/tmp
O=gpurun_out/r6_full
mkdir -p $O
timeout 2400 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -6 > $O/pytest.txt; tail -3 $O/pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 1500 python bench.py --steps 20 --warmup 5 2>$O/bench_20.err | tail -1 > $O/bench_20.json
python -c "import json; d=json.load(open('$O/bench_20.json')); print(round(d['value']), round(d['ms_per_step'],4), d['roofline']['frac'], list(d.get('secondary',{}).keys()))"
