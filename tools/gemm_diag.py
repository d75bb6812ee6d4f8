import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogstream_amd import ops
dev = torch.device("cuda:0")
def t(M, N, K):
    a = (torch.rand(M, K, device=dev) * 2 - 1).bfloat16(); w = ((torch.rand(N, K, device=dev) * 2 - 1) * .05).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ts = []
    for i in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.gemm(a, w, out=out); e1.record(); torch.cuda.synchronize()
        if i: ts.append(e0.elapsed_time(e1))
    ts.sort(); print(f"M{M} N{N} K{K}: {ts[len(ts)//2]:.3f} ms  {2.0*M*N*K/ts[len(ts)//2]/1e9:.0f} TF")
for shape in [(59136, 4352, 1152), (59136, 4352, 2304), (59136, 4352, 4608), (59136, 3584, 1152), (65536, 4096, 1152), (65536, 4096, 4096)]:
    t(*shape)
