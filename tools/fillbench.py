import torch
dev=torch.device("cuda")
x=torch.empty(256*1024*32*131, device=dev); y=torch.empty_like(x)
for name,fn in [("fill", lambda: x.fill_(1.0)), ("copy", lambda: y.copy_(x))]:
    for _ in range(2): fn()
    ts=[]
    for _ in range(5):
        a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    print(name, min(ts), "ms", x.numel()*4/min(ts)/1e6, "GB/s (one-sided bytes)")
