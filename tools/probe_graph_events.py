"""GPU probe: can HIP events recorded INSIDE a captured graph (torch.cuda.Event(external=True) -> hipEventRecordWithFlags
external) be read after a replay?  If so bench.py can time the GEMM family inside the replayed step itself."""
import torch
dev = torch.device("cuda:0")
a = torch.randn(4096, 4096, device=dev)
b = torch.randn(4096, 4096, device=dev)
c = torch.empty_like(a)
s = torch.cuda.Stream()
evs = [torch.cuda.Event(enable_timing=True, external=True) for _ in range(4)]
torch.cuda.synchronize()
with torch.cuda.stream(s):
    torch.mm(a, b, out=c)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        evs[0].record()
        torch.mm(a, b, out=c)
        evs[1].record()
        c.mul_(2.0)
        evs[2].record()
        torch.mm(a, b, out=c)
        evs[3].record()
    for _ in range(3):
        g.replay()
torch.cuda.synchronize()
print("in-graph event times (ms):", [evs[i].elapsed_time(evs[i + 1]) for i in range(3)])
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); torch.mm(a, b, out=c); e1.record(); torch.cuda.synchronize()
print("eager mm (ms):", e0.elapsed_time(e1))
