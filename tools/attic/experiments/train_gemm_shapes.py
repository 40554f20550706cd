"""Which GEMMs / reductions of one training step cost what: torch.profiler with input shapes, grouped by (op, shapes)."""
import importlib.util, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import ProfilerActivity, profile
spec = importlib.util.spec_from_file_location("tsb", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "train_step_bench.py"))
tsb = importlib.util.module_from_spec(spec); spec.loader.exec_module(tsb)
dev = torch.device("cuda")
feat, xyz, label = tsb.make_scan(24000, 3, dev)
torch.manual_seed(0)
net = tsb.FirstStage().to(dev).train()
opt = torch.optim.Adam(net.parameters(), lr=1e-3, fused=True)
def step():
    offset, sem = net(feat)
    loss, _ = tsb.losses(offset, sem, xyz, label)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()
for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key in ("aten::mm", "aten::addmm", "aten::bmm", "aten::sum", "aten::add", "aten::sub", "aten::mul", "aten::neg", "aten::copy_", "aten::index_add_", "aten::cat")]
rows.sort(key=lambda e: -e.device_time_total)
for e in rows[:40]:
    print(f"{e.key:18s} {e.device_time_total / 1e3:8.3f} ms  {e.count:4d} calls  {str(e.input_shapes)[:150]}")
