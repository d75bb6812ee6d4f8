"""Seeded inputs of the k-means near-tie study, shared by tests/golden/kmeans_tie_study.py (reference side, build
container) and the GPU parity test (both regenerate them; only outputs + a checksum are stored)."""
import torch

K, P, D = 11, 50, 3584          # K = ceil(160 / 15); P*D = the real cfg3 width (50 tokens x 3584)
T = 400


def tie_inputs():
    g = torch.Generator().manual_seed(20250824)
    PD = P * D
    base = torch.randn(PD, generator=g)                         # what all frames of a clip share
    centres = base[None] + 0.1 * torch.randn(K, PD, generator=g)
    n = T - K
    a = torch.randint(0, K, (n,), generator=g)
    b = (a + 1 + torch.randint(0, K - 1, (n,), generator=g)) % K
    delta = 10 ** (-9 + 6.5 * torch.rand(n, generator=g))       # log-uniform in [1e-9, 3e-3]
    x = torch.empty(T, PD)
    x[:K] = centres
    for i in range(n):
        ca, cb = centres[a[i]], centres[b[i]]
        x[K + i] = ca + (0.5 - delta[i]) * (cb - ca) + 0.002 * torch.randn(PD, generator=g)
    return {"features": x.view(T, P, D), "pair": torch.stack([a, b], 1), "delta": delta}
