#!/usr/bin/env python3
"""k-means near-tie study (SURVEY.md section 7 "Hard parts"; build container only: it RUNS the reference).

torch.cdist takes the matmul form sqrt(clamp(|x|^2 + |c|^2 - 2 x.c, 0)) in ONE fp32 sgemm whenever either operand
has more than 25 rows (kmeans_with_time.py:48,73 -- always, since k-means only runs for T >= 136). At the real
width (P*D = 50 x 3584 = 179 200) |x|^2 ~ 1e5 while a row's two candidate distances may differ by far less, so the
reference's argmin near a cluster boundary is decided by the rounding of its own sgemm. The HIP kernel sums
(x - c)^2 directly (no cancellation). This script measures where the two part company:

  * plants K centres and rows on the segment between two of them at a controlled relative margin
    (d_far - d_near) / d_near from 1e-2 down to 1e-8, all timestamps equal (the time term is then zero for every
    cluster, kmeans_with_time.py:96-99, and the assignment is the argmin of the feature distance alone);
  * runs THE REFERENCE's kmeans_with_time_min_max for one Lloyd iteration from the planted centres (the RNG calls
    of its k-means++ are answered with the planted rows), with 8 threads and with 1 thread;
  * computes the exact (fp64) assignment and the one the HIP arithmetic gives (restated below: fp32 squares,
    512-column slices, fp64 combination -- what csrc/kmeans.hip did in round 1 -- and full fp64 accumulation,
    what it does now);
  * writes tests/golden/kmeans_ties.npz: seeds, per-row margin, the reference's assignments, and the flip table.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/kmeans_tie_study.py"""
import os
import random
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True

from model import kmeans_with_time as ref_mod  # noqa: E402

from tie_inputs import K, P, D, tie_inputs  # noqa: E402


def ref_assign(feats, threads):
    """one Lloyd iteration of the reference from the planted centres (rows 0..K-1)"""
    torch.set_num_threads(threads)
    T = feats.shape[0]
    ts = torch.zeros(T)
    picks = iter(range(1, K))
    orig_randint, orig_multinomial = random.randint, torch.multinomial
    random.randint = lambda a, b: 0
    torch.multinomial = lambda probs, n, **kw: torch.tensor([next(picks)])
    try:
        _, _, assign = ref_mod.kmeans_with_time_min_max(feats, ts, K, max_iteration=1)
    finally:
        random.randint, torch.multinomial = orig_randint, orig_multinomial
    return assign


def exact_d2(x, c):
    out = torch.empty(x.shape[0], c.shape[0], dtype=torch.float64)
    for k in range(c.shape[0]):
        out[:, k] = (x.double() - c[k].double()).pow(2).sum(1)
    return out


def direct_fp32_d2(x, c, sl=512):
    """round-1 HIP arithmetic: fp32 sum of squares per 512-column slice, slices combined in fp64"""
    T, PD = x.shape
    out = torch.zeros(T, c.shape[0], dtype=torch.float64)
    for k in range(c.shape[0]):
        d = (x - c[k]).pow(2).view(T, PD // sl, sl).sum(2, dtype=torch.float32)
        out[:, k] = d.double().sum(1)
    return out.float()


def main():
    inp = tie_inputs()
    x, pair, delta = inp["features"], inp["pair"], inp["delta"]
    T = x.shape[0]
    flat = x.view(T, P * D)
    c = flat[:K]
    d2 = exact_d2(flat, c)
    srt = d2.sqrt().sort(dim=1).values
    margin = ((srt[:, 1] - srt[:, 0]) / srt[:, 0].clamp_min(1e-30)).numpy()         # true relative margin, fp64
    exact = d2.argmin(1)
    a8 = ref_assign(x, 8)
    a1 = ref_assign(x, 1)
    d32 = direct_fp32_d2(flat, c).argmin(1)
    d64 = exact_d2(flat, c).float().argmin(1)                                        # fp64 accumulation, rounded once
    rows = np.arange(T) >= K
    edges = [1e-2, 1e-3, 1e-4, 1e-5, 1e-6, 1e-7, 1e-8, 0.0]
    table = []
    print(f"{'margin bin':>18} {'rows':>5} {'ref8!=exact':>11} {'ref1!=exact':>11} {'ref8!=ref1':>10} {'hip32!=exact':>12} {'hip64!=exact':>12}")
    hi = np.inf
    for lo in edges:
        m = rows & (margin < hi) & (margin >= lo)
        r = [int(m.sum()), int((a8.numpy() != exact.numpy())[m].sum()), int((a1.numpy() != exact.numpy())[m].sum()),
             int((a8.numpy() != a1.numpy())[m].sum()), int((d32.numpy() != exact.numpy())[m].sum()),
             int((d64.numpy() != exact.numpy())[m].sum())]
        table.append([hi if np.isfinite(hi) else 1.0, lo] + r)
        print(f"[{lo:8.0e},{hi:8.0e}) {r[0]:5d} {r[1]:11d} {r[2]:11d} {r[3]:10d} {r[4]:12d} {r[5]:12d}")
        hi = lo
    norm2 = float(flat.double().pow(2).sum(1).mean())
    dist2 = float(srt[rows.nonzero()[0], 0].pow(2).mean())
    print(f"mean |x|^2 = {norm2:.4g}, mean nearest d^2 = {dist2:.4g}: one fp32 ulp of |x|^2 is {norm2 * 2**-24:.3g}, "
          f"i.e. {norm2 * 2**-24 / dist2:.2e} of d^2")
    np.savez_compressed(os.path.join(HERE, "kmeans_ties.npz"), margin=margin, ref_assign_8t=a8.numpy(), ref_assign_1t=a1.numpy(),
                        exact_assign=exact.numpy(), table=np.array(table), pair=pair.numpy(), delta=delta.numpy(),
                        checksum=np.float64(flat.double().abs().sum()), mean_norm2=np.float64(norm2), mean_d2=np.float64(dist2))
    print("wrote kmeans_ties.npz")


if __name__ == "__main__":
    main()
