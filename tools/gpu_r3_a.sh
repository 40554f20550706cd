#!/bin/bash
# round 3, call 1: smoke, all GPU tests (new whole-net parity), headline bench, two ranks sharing the one GPU (gloo).
set -u
mkdir -p gpurun_out/r3a
export TMPDIR=/tmp
O=gpurun_out/r3a
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "== pytest gpu"; timeout 1200 python -m pytest tests -m gpu -q -rf -s 2>&1 | tee $O/pytest.log | grep -v "^\s*$" | tail -45
echo "== bench"; timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json
echo "== bench 2 ranks on one GPU (gloo)"; timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 10 --warmup 2 --backend gloo > $O/bench_2ranks.json 2> $O/bench_2ranks.err; tail -c 1200 $O/bench_2ranks.json; tail -3 $O/bench_2ranks.err
