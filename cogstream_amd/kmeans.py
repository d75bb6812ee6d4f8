"""Time-aware k-means with the reference's call surface, device steps through the C ABI.

    kmeans_with_time_min_max(features, timestamp, cluster_num, alpha=2, max_iteration=30, tol=1e-4)
        -> (centres [K,P,D] fp32, centre_times [K], assignments [T] int64)     model/kmeans_with_time.py:4-137
    select_additional_frames(cls_feature, long_memory, cluster_assignments, n)  model/cogreasoner_chat.py:50-64

The random draws stay on the host exactly where the reference makes them (python `random` for the first
centre and for empty-cluster reseeds, torch.multinomial on the CPU generator for k-means++), so that under
the same seeds the same indices are drawn; the feature arithmetic (distances, assignments, means, centre
shift) runs in HIP kernels over the features resident in HBM (bf16 or fp32, never copied to the host)."""
from __future__ import annotations

import random
from typing import List, Optional, Tuple

import torch

from . import ops

last_stats = {"kpp_passes": 0, "iterations": 0, "kpp_path": "", "min_rel_margin": float("inf"), "rows_below_1e-3": 0}
"""of the most recent call (bench.py: bytes moved per stage; DESIGN.md section 2: how close the assignments were).
min_rel_margin = over all rows and Lloyd iterations, the smallest relative change of a row's feature distances that
would flip its cluster (cogs_kmeans_margins; (d2 - d1) / mean(d1, d2) of the two nearest centres when the time terms are
equal); rows_below_1e-3 = (row, iteration) pairs under 1e-3, where the reference's own cdist rounding decides a row."""


def _lloyd_from_rows(x, ts, rows, K: int, PD: int, alpha: float, max_iteration: int, tol: float, ws):
    """Lloyd iterations (:71-131) from the centres x[rows], inside the library. Empty clusters are reseeded with
    random.randint(0, T-1) per cluster in ascending order (:116-120): the values are drawn AHEAD into a pool the device
    consumes in that order, and the generator is then put back to where the reference would have left it (state
    restored, the used number of draws replayed). -> (centres [K, PD] fp32, centre_ts, assign, iterations)"""
    T = x.shape[0]
    centres = ops.pack_rows(x.index_select(0, rows), torch.float32, PD) if x.dtype != torch.float32 \
        else x.index_select(0, rows).contiguous()
    centre_ts = ts.index_select(0, rows).contiguous()
    assign = torch.empty(T, dtype=torch.int64, device=x.device)
    state = random.getstate()
    left, done, used_total, pool_n = max_iteration, 0, 0, 2 * K
    drawn: List[int] = []
    while left > 0:
        drawn.extend(random.randint(0, T - 1) for _ in range(pool_n))
        it, used, exhausted = ops.kmeans_lloyd(x, ts, centres, centre_ts, assign, float(alpha), left, float(tol),
                                               drawn[used_total:], ws)
        done, left, used_total = done + it, left - it, used_total + used
        if not exhausted:
            break
        pool_n = min(4096, max(pool_n * 2, K))     # an iteration wanted more reseeds than were left: draw more, go on
        drawn = drawn[:used_total]
        random.setstate(state)
        for _ in range(used_total):
            random.randint(0, T - 1)
    random.setstate(state)
    for _ in range(used_total):
        random.randint(0, T - 1)
    return centres, centre_ts, assign, done


def kmeans_with_time_min_max(features: torch.Tensor, timestamp, cluster_num: int, alpha: float = 2,
                             max_iteration: int = 30, tol: float = 1e-4
                             ) -> Tuple[torch.Tensor, torch.Tensor, Optional[torch.Tensor]]:
    if not isinstance(timestamp, torch.Tensor):
        timestamp = torch.tensor(timestamp, dtype=torch.float32)
    T, P, D = features.shape
    if T <= cluster_num:  # kmeans_with_time.py:30-32
        return features.to(torch.float32), timestamp[:cluster_num], None
    dev = features.device
    x = features.reshape(T, P * D).contiguous()
    ts = timestamp.to(dev, torch.float32).contiguous()
    K, PD = cluster_num, P * D
    ws = ops.kmeans_workspace(T, PD, K, dev)

    # ---- k-means++ on feature distance only (:41-62). The reference draws centre m with torch.multinomial(probs, 1) on
    # the CPU generator, which IS argmax(probs / q) with q = empty_like(probs).exponential_(1) (ATen, n_sample == 1). The
    # K - 1 rows of q are drawn here, exactly as those calls would consume the generator, and the whole seeding runs in
    # one library call (cogs_kmeans_pp) with no host round trip per centre. If a step finds all probabilities zero --
    # the reference then calls random.randint instead of multinomial -- both generators are put back and the seeding is
    # redone step by step (the round-3 path: one pinned read of the [T] distances per centre) ----
    first = random.randint(0, T - 1)
    py_state, torch_state = random.getstate(), torch.get_rng_state()
    q = torch.stack([torch.empty(T, dtype=torch.float32).exponential_(1) for _ in range(K - 1)])
    idx_d, flag = ops.kmeans_pp(x, first, K, q.to(dev, non_blocking=True), ws)
    # (no host read here: the Lloyd loop below starts from the device-resident rows; the flag is looked at after its
    # first synchronisation, and in the rare flagged case everything is redone step by step)
    run = _lloyd_from_rows(x, ts, idx_d.to(torch.int64), K, PD, alpha, max_iteration, tol, ws)
    if int(flag.item()) == 0:
        last_stats["kpp_path"] = "one call"
    else:
        random.setstate(py_state)
        torch.set_rng_state(torch_state)
        idx: List[int] = [first]
        nearest2 = torch.empty(T, dtype=torch.float32, device=dev)
        probs = torch.empty(T, dtype=torch.float32).pin_memory()
        while len(idx) < K:
            ops.kmeans_pp_step(x, idx[-1], len(idx) == 1, nearest2, probs, ws)
            probs.sqrt_().square_()                        # the reference's (cdist distance) ** 2 (:53)
            s = probs.sum()
            if s.item() == 0:
                new = random.randint(0, T - 1)
            else:
                new = int(torch.multinomial(probs / s, 1).item())        # CPU generator, like the reference
            idx.append(new)
        last_stats["kpp_path"] = "step by step (a zero-probability step)"
        run = _lloyd_from_rows(x, ts, torch.tensor(idx, dtype=torch.int64, device=dev), K, PD, alpha, max_iteration, tol, ws)
    centres, centre_ts, assign, done = run
    last_stats["kpp_passes"], last_stats["iterations"] = K - 1, done
    last_stats["min_rel_margin"], last_stats["rows_below_1e-3"] = ops.kmeans_margins(T, PD, K, ws)
    return centres.view(K, P, D), centre_ts, assign


def select_additional_frames(cls_feature: torch.Tensor, long_memory: torch.Tensor,
                             cluster_assignments: torch.Tensor, additional_frame_num: int) -> List[torch.Tensor]:
    """per cluster: all members if <= n, else the n members nearest (L2 over P*D) to the centroid. Distances and the
    picks are computed on the device (cogs_kmeans_sqdist + cogs_select_near_centroid); only the K counts cross."""
    T = cls_feature.shape[0]
    K = long_memory.shape[0]
    dev = cls_feature.device
    x = cls_feature.reshape(T, -1).contiguous()
    c = long_memory.reshape(K, -1).to(torch.float32).contiguous()
    ws = ops.kmeans_workspace(T, x.shape[1], K, dev)
    d2 = ops.kmeans_sqdist(x, c, None, K, ws)
    assign = cluster_assignments.to(dev, torch.int64).contiguous()
    if additional_frame_num > 8:                       # beyond the kernel's per-lane list: host fallback of round 1
        d2h, ah = d2.cpu(), assign.cpu()
        out = []
        for i in range(K):
            members = torch.nonzero(ah == i, as_tuple=True)[0]
            if members.numel() <= additional_frame_num:
                out.append(members.to(dev))
            else:
                _, top = torch.topk(d2h[members, i], k=additional_frame_num, largest=False)
                out.append(members[top].to(dev))
        return out
    picks, counts = ops.select_near_centroid(d2, assign, additional_frame_num)
    return [picks[i, :n] for i, n in enumerate(counts.tolist())]
