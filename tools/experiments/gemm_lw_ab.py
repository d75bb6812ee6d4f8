#!/usr/bin/env python3
"""A/B of the loader-wave GEMM prototype (tools/experiments/gemm_lw.hip, cogs_x_gemm_lw) against the shipped cogs_gemm on the four
ViT shapes of cfg2: python tools/experiments/gemm_lw_ab.py [M]. Correctness first (against torch fp32 matmul of the bf16 operands,
ragged M / N), then back-to-back timings. --nostore on the command line puts cogs_gemm in its K-loop-only
diagnostic mode; the prototype's own nostore flag is timed beside it."""
import ctypes as C
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cogstream_amd import _lib as L, ops

# the prototype is NOT part of libcogs_hip.so (it lost): built here into its own library
HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libgemm_lw.so")
if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(os.path.join(HERE, "gemm_lw.hip")):
    import subprocess
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950",
                           "-I", os.path.join(ROOT, "cogstream_amd", "csrc"), os.path.join(HERE, "gemm_lw.hip"), "-o", SO])
dev = torch.device("cuda:0")
fn = C.CDLL(SO).cogs_x_gemm_lw
fn.restype = C.c_int
fn.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]


def lw(a, w, out, group_m=8, nostore=0):
    M, K = a.shape
    N = w.shape[0]
    rc = fn(L.current_stream(), a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), out.data_ptr(), out.stride(0), M, N, K,
            group_m, nostore)
    assert rc == 0, rc


def bench(f, n=30):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


# ---- correctness (ragged edges included)
for (M, N, K) in ((1000, 1152, 1152), (2048, 3456, 1152), (777, 200, 64), (3000, 1152, 4352)):
    a = (torch.rand(M, K, device=dev) * 2 - 1).bfloat16()
    w = ((torch.rand(N, K, device=dev) * 2 - 1) * 0.05).bfloat16()
    out = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    lw(a, w, out)
    ref = a.float() @ w.float().t()
    same = torch.equal(out, ops.gemm(a, w))
    err = float((out.float() - ref).abs().max() / ref.abs().max())
    print(f"check {M}x{N}x{K}: rel err {err:.2e}, bit-equal to cogs_gemm: {same}", flush=True)
    assert err < 1e-2

M = int(sys.argv[1]) if len(sys.argv) > 1 else 59136
nostore_env = "--nostore" in sys.argv
if nostore_env:
    from cogstream_amd import _lib as _L
    _L.debug_set("gemm_nostore", 1)
for name, N, K in (("o", 1152, 1152), ("fc2", 1152, 4352), ("qkv", 3456, 1152), ("fc1", 4352, 1152)):
    a = (torch.rand(M, K, device=dev) * 2 - 1).bfloat16()
    w = ((torch.rand(N, K, device=dev) * 2 - 1) * 0.05).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    fl = 2.0 * M * N * K
    t_ref = bench(lambda: ops.gemm(a, w, out=out))
    res = [f"cogs_gemm{'(nostore)' if nostore_env else ''} {t_ref * 1e3:.3f} ms {fl / t_ref / 1e12:.0f} TF"]
    for gm in (8, 2, 4):
        t = bench(lambda: lw(a, w, out, gm, 0))
        res.append(f"lw gm{gm} {t * 1e3:.3f} ms {fl / t / 1e12:.0f} TF")
    t = bench(lambda: lw(a, w, out, 8, 1))
    res.append(f"lw nostore {t * 1e3:.3f} ms {fl / t / 1e12:.0f} TF")
    print(f"{name} {M}x{N}x{K}: " + " | ".join(res), flush=True)
