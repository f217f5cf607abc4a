"""Does hipGraph replay run independent branches concurrently here?  Two chains of small kernels, serial vs forked."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dev = torch.device("cuda:0")
a = [torch.randn(154, 768, device=dev) for _ in range(2)]
w = [torch.randn(768, 768, device=dev) for _ in range(2)]

def chain(i, n=40):
    x = a[i]
    for _ in range(n):
        x = torch.tanh(x @ w[i]) * 0.5
    return x

def timeit(g):
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 5

chain(0); chain(1); torch.cuda.synchronize()
g1 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g1):
    r0 = chain(0); r1 = chain(1)
side = torch.cuda.Stream()
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2):
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        s1 = chain(1)
    s0 = chain(0)
    cur.wait_stream(side)
print("serial graph  : %.3f ms" % timeit(g1))
print("forked graph  : %.3f ms" % timeit(g2))
# eager two streams
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    chain(1)
chain(0)
torch.cuda.current_stream().wait_stream(side)
e1.record(); torch.cuda.synchronize()
print("eager 2 streams: %.3f ms" % e0.elapsed_time(e1))
