"""Frame sharding of the encoder over the GPUs of one node (build-only, the reference
never shards a clip). Legal because block-diagonal attention, RoPE and the 2x2 merge are all per frame
(model/modeling_videollama3_encoder.py:309-312,427,487-501): rank r encodes AND PROJECTS a contiguous slice of
frames, one all-gather (RCCL over xGMI on the GPU, gloo in the CPU tests; `cogs_allgather_tokens` is the C-ABI form)
reassembles the projected [M, 3584] visual tokens in frame order. One process per GPU; no other data-path collective.

Why the gather carries the 3584-wide PROJECTED tokens and not the 1152-wide encoder output (SURVEY 8e budgets the
narrow form: 29.5 MB instead of 91.7 MB at cfg3): the consumer needs projected tokens, so the choice is between
  narrow gather + projector on all M tokens at the consumer:  3.7 MB per link (~0.04 ms at ~100 GB/s per xGMI link, the
      7 shards arrive over 7 links in parallel) + 0.41 ms (the two projector GEMMs at M = 12 800, rocprofv3), and
  projector on the rank's own M/8 tokens + wide gather:        0.13 ms (M = 1 600) + 11.5 MB per link (~0.12 ms).
The wide form is ~0.2 ms shorter per clip and does the projector's 0.43 TFLOP once instead of on every consumer; either
way the collective is < 4 % of a rank's 5.9 ms share of the clip. (Link rate is the guide's 7 x ~153 GB/s peak derated
to ~100 GB/s; no multi-GPU node was available to the builder, so these are estimates -- the driver's N = 2/4/8 runs of
bench.py measure the real thing.)"""
from __future__ import annotations

from typing import List, Tuple

import torch
import torch.distributed as dist


def frame_shards(num_frames: int, world: int) -> List[Tuple[int, int]]:
    """contiguous [begin, end) frame ranges, sizes differing by at most one"""
    base, rem = divmod(num_frames, world)
    out, b = [], 0
    for r in range(world):
        e = b + base + (1 if r < rem else 0)
        out.append((b, e))
        b = e
    return out


def shard_video(pixel_values: torch.Tensor, grid_size: Tuple[int, int, int], rank: int, world: int):
    """rows of this rank's frames of ONE video + its local grid"""
    t, gh, gw = (int(v) for v in grid_size)
    b, e = frame_shards(t, world)[rank]
    per = gh * gw
    return pixel_values[b * per:e * per], torch.tensor([[e - b, gh, gw]])


def gather_tokens(local_tokens: torch.Tensor, grid_size: Tuple[int, int, int], merge_size: int, world: int,
                  group=None) -> torch.Tensor:
    """all-gather the per-rank merged tokens back into frame order -> [M, hidden] on every rank"""
    if world == 1:
        return local_tokens
    t, gh, gw = (int(v) for v in grid_size)
    ppf = (gh // merge_size) * (gw // merge_size)
    shards = frame_shards(t, world)
    if local_tokens.is_cuda and dist.get_backend(group) == "gloo":
        # gloo has no device all-gather: stage through host memory (CPU tests and one-GPU rehearsals only;
        # on a node the backend is nccl = RCCL and the tensors never leave HBM / xGMI)
        return gather_tokens(local_tokens.cpu(), grid_size, merge_size, world, group).to(local_tokens.device)
    if all(e - b == shards[0][1] - shards[0][0] for b, e in shards):
        out = torch.empty(t * ppf, local_tokens.shape[1], dtype=local_tokens.dtype, device=local_tokens.device)
        dist.all_gather_into_tensor(out, local_tokens.contiguous(), group=group)
        return out
    # ragged split: collectives need equal counts, so pad every shard to the largest one and trim after
    rows = max(e - b for b, e in shards) * ppf
    padded = local_tokens.new_zeros(rows, local_tokens.shape[1])
    padded[: local_tokens.shape[0]] = local_tokens
    out = torch.empty(world * rows, local_tokens.shape[1], dtype=local_tokens.dtype, device=local_tokens.device)
    dist.all_gather_into_tensor(out, padded, group=group)
    return torch.cat([out[r * rows: r * rows + (e - b) * ppf] for r, (b, e) in enumerate(shards)], dim=0)


def partition_sequences(lengths, world: int) -> List[List[int]]:
    """split independent sequences over `world` ranks, balancing tokens (prefill cost): longest first onto the
    least-loaded rank, ties to the lower rank / lower index, so every rank derives the same plan"""
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    load = [0] * world
    plan: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda j: (load[j], j))
        plan[r].append(i)
        load[r] += int(lengths[i])
    return [sorted(p) for p in plan]


def pooled_means_sharded(forward_segments, segments: List[torch.Tensor], rank: int, world: int, group=None) -> torch.Tensor:
    """Event-summary passes of select_events_based_on_summary (model/cogreasoner_chat.py:303-322) over the ranks
    (SURVEY.md section 8f rank 3): the K event prompts and the question are independent sequences, so rank r runs ONE
    var-len forward over its share and a single all-gather of the [*, H] fp32 means follows (K*H*4 bytes: 258 KB at
    K = 18). `segments`: per-sequence input embeddings [n_i, H], identical on every rank;
    `forward_segments(cat, lens) -> [len(lens), H] fp32`. Returns [len(segments), H] fp32 in sequence order on every
    rank, equal to forward_segments over all of them (sequences do not interact)."""
    lens = [int(x.shape[0]) for x in segments]
    if world == 1:
        return forward_segments(torch.cat(segments), lens)
    plan = partition_sequences(lens, world)
    mine = plan[rank]
    H = segments[0].shape[1]
    dev = segments[0].device
    width = max(len(p) for p in plan)
    local = torch.zeros(width, H, dtype=torch.float32, device=dev)
    if mine:
        local[: len(mine)] = forward_segments(torch.cat([segments[i] for i in mine]), [lens[i] for i in mine])
    if local.is_cuda and dist.get_backend(group) == "gloo":   # one-GPU rehearsal: see gather_tokens
        stage = local.cpu()
        out = torch.empty(world * width, H, dtype=torch.float32)
        dist.all_gather_into_tensor(out, stage, group=group)
        out = out.to(dev)
    else:
        out = torch.empty(world * width, H, dtype=torch.float32, device=dev)
        dist.all_gather_into_tensor(out, local, group=group)
    res = torch.empty(len(segments), H, dtype=torch.float32, device=dev)
    for r, p in enumerate(plan):
        if p:
            res[torch.tensor(p, device=dev)] = out[r * width: r * width + len(p)]
    return res
