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

from typing import List, Sequence, Tuple

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


def _world_gather(local: torch.Tensor, out: torch.Tensor, group=None) -> None:
    """all_gather_into_tensor, staged through host memory when the tensors are on a GPU but the group is gloo (CPU
    tests and one-GPU rehearsals only; on a node the backend is nccl = RCCL and nothing leaves HBM / xGMI)"""
    if local.is_cuda and dist.get_backend(group) == "gloo":
        host = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(host, local.cpu().contiguous(), group=group)
        out.copy_(host)
    else:
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)


def gather_rows(local: torch.Tensor, counts: Sequence[int], group=None) -> torch.Tensor:
    """all-gather row blocks of known sizes (`counts[r]` rows on rank r, every rank knows all of them) -> the
    concatenation in rank order, on every rank. Equal counts: one collective straight into the result. Ragged:
    collectives need equal counts, so every block is padded to the largest one and the padding trimmed after."""
    world = len(counts)
    if world == 1:
        return local
    width = local.shape[1]
    rows = max(int(c) for c in counts)
    if all(int(c) == rows for c in counts):
        out = torch.empty(world * rows, width, dtype=local.dtype, device=local.device)
        _world_gather(local, out, group)
        return out
    padded = local.new_zeros(rows, width)
    padded[: local.shape[0]] = local
    out = torch.empty(world * rows, width, dtype=local.dtype, device=local.device)
    _world_gather(padded, out, group)
    return torch.cat([out[r * rows: r * rows + int(c)] for r, c in enumerate(counts) if int(c) > 0], dim=0)


def gather_tokens(local_tokens: torch.Tensor, grid_size: Tuple[int, int, int], merge_size: int, world: int,
                  group=None) -> torch.Tensor:
    """all-gather the per-rank merged tokens of ONE video back into frame order -> [M, hidden] on every rank"""
    if world == 1:
        return local_tokens
    t, gh, gw = (int(v) for v in grid_size)
    ppf = (gh // merge_size) * (gw // merge_size)
    return gather_rows(local_tokens, [(e - b) * ppf for b, e in frame_shards(t, world)], group)


class FramePlan:
    """Which frames of a request (several videos, each t x gh x gw patches) a rank encodes.
    The frames of all videos in order form one sequence; it is cut into `world` CONTIGUOUS runs whose patch counts
    are as even as frame boundaries allow (cut r sits at the frame boundary nearest to r/world of the patches), so
    the gathered tokens are already in the reference's order (video -> frame -> token). Every rank derives the same
    plan from grid_sizes / merge_sizes alone.

      pieces[r]       [(video, first frame, end frame), ...] of rank r (empty when there are more ranks than frames)
      patch_rows[r]   (begin, end) rows of pixel_values          token_counts[r]   merged tokens rank r produces"""

    def __init__(self, grid_sizes, merge_sizes, world: int):
        gs = [[int(x) for x in g] for g in (grid_sizes.tolist() if hasattr(grid_sizes, "tolist") else grid_sizes)]
        ms = [int(m) for m in (merge_sizes.tolist() if hasattr(merge_sizes, "tolist") else merge_sizes)]
        frames = [(v, f, gh * gw, (gh // m) * (gw // m)) for v, ((t, gh, gw), m) in enumerate(zip(gs, ms)) for f in range(t)]
        cum = [0]
        for _, _, n, _ in frames:
            cum.append(cum[-1] + n)
        total = cum[-1]
        cuts = [0]
        for r in range(1, world):
            target = total * r / world
            # nearest frame boundary at or after the previous cut (ties to the earlier boundary)
            j = min(range(cuts[-1], len(cum)), key=lambda i: (abs(cum[i] - target), i))
            cuts.append(j)
        cuts.append(len(frames))
        self.world, self.grids, self.merges = world, gs, ms
        self.pieces: List[List[Tuple[int, int, int]]] = []
        self.patch_rows: List[Tuple[int, int]] = []
        self.token_counts: List[int] = []
        for r in range(world):
            run = frames[cuts[r]:cuts[r + 1]]
            pcs: List[Tuple[int, int, int]] = []
            for v, f, _, _ in run:
                if pcs and pcs[-1][0] == v and pcs[-1][2] == f:
                    pcs[-1] = (v, pcs[-1][1], f + 1)
                else:
                    pcs.append((v, f, f + 1))
            self.pieces.append(pcs)
            self.patch_rows.append((cum[cuts[r]], cum[cuts[r + 1]]))
            self.token_counts.append(sum(m for _, _, _, m in run))

    def local(self, pixel_values: torch.Tensor, rank: int):
        """-> (rows of this rank's frames (a view), grid_sizes [n,3], merge_sizes [n]) for the encoder call"""
        b, e = self.patch_rows[rank]
        pcs = self.pieces[rank]
        grid = torch.tensor([[fe - fb, self.grids[v][1], self.grids[v][2]] for v, fb, fe in pcs], dtype=torch.int64).reshape(-1, 3)
        merge = torch.tensor([self.merges[v] for v, _, _ in pcs], dtype=torch.int64)
        return pixel_values[b:e], grid, merge


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
