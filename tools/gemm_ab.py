import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogstream_amd import ops
dev = torch.device("cuda:0")
def t(M, N, K, res):
    a = (torch.rand(M, K, device=dev) * 2 - 1).bfloat16(); w = ((torch.rand(N, K, device=dev) * 2 - 1) * .05).bfloat16()
    r = torch.rand(M, N, device=dev).bfloat16() if res else None
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ts = []
    for i in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.gemm(a, w, out=out, residual=r); e1.record(); torch.cuda.synchronize()
        if i: ts.append(e0.elapsed_time(e1))
    ts.sort(); print(f"M{M} N{N} K{K} res={int(res)}: {ts[len(ts)//2]:.3f} ms  {2.0*M*N*K/ts[len(ts)//2]/1e9:.0f} TF")
for (M, N, K) in [(15396, 3584, 3584), (15360, 3584, 3584), (15396, 4608, 3584), (15396, 3456, 3584), (59136, 3584, 1152)]:
    for res in (False, True):
        t(M, N, K, res)
