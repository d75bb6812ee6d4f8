"""Oracle: time-aware k-means and near-centroid frame pick (model/kmeans_with_time.py:4-137,
model/cogreasoner_chat.py:50-64). TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py."""
from __future__ import annotations

import random

import torch


def kmeans_with_time_min_max(features, timestamp, cluster_num, alpha=2, max_iteration=30, tol=1e-4,
                             exact_distances: bool = False):
    """Restates kmeans_with_time.py:4-137 step by step; RNG: python `random` for the first centre and
    reseeds (:41,57,118), torch.multinomial on the CPU generator for the rest (:60).
    exact_distances (tie studies only, not the reference's arithmetic): torch.cdist without its matmul form, i.e. the
    direct sum of squared differences. With more than 25 rows the reference's cdist evaluates |x|^2 + |c|^2 - 2 x.c in
    one sgemm, whose rounding noise breaks EXACT ties between duplicate centres in an order no other implementation can
    reproduce (tests/golden/kmeans_tie_study.py); the direct form leaves exact ties exact, and
    argmin then takes the first centre -- as the HIP kernels do."""
    cdist = (lambda a, b: torch.cdist(a, b, p=2, compute_mode="donot_use_mm_for_euclid_dist")) if exact_distances \
        else (lambda a, b: torch.cdist(a, b, p=2))
    features = features.to(dtype=torch.float32)
    if not isinstance(timestamp, torch.Tensor):
        timestamp = torch.tensor(timestamp, dtype=torch.float32)
    T, P, D = features.shape
    if T <= cluster_num:
        return features, timestamp[:cluster_num], None
    x = features.reshape(T, P * D)
    idx = [random.randint(0, T - 1)]
    while len(idx) < cluster_num:
        d = cdist(x, x[idx])
        nearest, _ = d.min(dim=1)
        probs = nearest ** 2
        s = probs.sum()
        if s.item() == 0:
            new = random.randint(0, T - 1)
        else:
            new = torch.multinomial(probs / s, 1).item()
        idx.append(new)
    cf, ct = x[idx], timestamp[idx]
    assign = None
    for _ in range(max_iteration):
        df = cdist(x, cf)
        dt = torch.abs(timestamp.unsqueeze(1) - ct.unsqueeze(0))
        fmin, fmax = df.min(dim=1, keepdim=True).values, df.max(dim=1, keepdim=True).values
        tmin, tmax = dt.min(dim=1, keepdim=True).values, dt.max(dim=1, keepdim=True).values
        nf = torch.where(fmax > fmin, (df - fmin) / (fmax - fmin), torch.zeros_like(df))
        nt = torch.where(tmax > tmin, (dt - tmin) / (tmax - tmin), torch.zeros_like(dt))
        final = torch.sqrt(nf ** 2 + alpha * (nt ** 2))
        assign = final.argmin(dim=1)
        ncf, nct = torch.zeros_like(cf), torch.zeros_like(ct)
        for i in range(cluster_num):
            m = assign == i
            if m.any():
                ncf[i] = x[m].mean(dim=0)
                nct[i] = timestamp[m].mean()
            else:
                r = random.randint(0, T - 1)
                ncf[i] = x[r]
                nct[i] = timestamp[r]
        shift = torch.norm(ncf - cf, p=2, dim=1).sum() + torch.norm(nct - ct, p=2).sum()
        cf, ct = ncf, nct
        if shift <= tol:
            break
    return cf.view(cluster_num, P, D), ct, assign


def select_additional_frames(cls_feature, long_memory, cluster_assignments, additional_frame_num):
    """cogreasoner_chat.py:50-64"""
    cls_feature = cls_feature.to(dtype=torch.float32)
    lm = long_memory.reshape(long_memory.shape[0], -1)
    flat = cls_feature.reshape(cls_feature.shape[0], -1)
    out = []
    for i in range(lm.shape[0]):
        mask = cluster_assignments == i
        feats = flat[mask]
        members = torch.nonzero(mask, as_tuple=True)[0]
        if feats.shape[0] <= additional_frame_num:
            out.append(members)
        else:
            d = torch.cdist(feats, lm[i].unsqueeze(0))
            _, top = torch.topk(d.squeeze(1), k=additional_frame_num, largest=False)
            out.append(members[top])
    return out
