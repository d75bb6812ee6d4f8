"""N > 1 path on CPU: two gloo ranks shard the frames of a clip, encode their slices (oracle stands in for the
HIP encoder: the collective logic is device-agnostic) and all-gather the visual tokens; the result must be
bit-identical to the single-process encode (block-diagonal attention makes frames independent)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

VIT = dict(hidden_size=144, intermediate_size=96, num_hidden_layers=1, num_attention_heads=2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, T, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from cogstream_amd.parallel import gather_tokens, shard_video
    from cogstream_amd.weights import VisionConfig, random_vit_state
    from oracle import vision as ov
    cfg = VisionConfig(**VIT)
    st = random_vit_state(cfg, seed=11, std=0.05)
    g = torch.Generator().manual_seed(3)
    grid = (T, 4, 4)
    pix = torch.rand(T * 16, 588, generator=g) * 2 - 1
    local_pix, local_grid = shard_video(pix, grid, rank, world)
    tok = ov.encode(st, local_pix, local_grid, torch.tensor([2]), heads=2, layers=1)
    full = gather_tokens(tok, grid, 2, world)
    if rank == 0:
        ref = ov.encode(st, pix, torch.tensor([list(grid)]), torch.tensor([2]), heads=2, layers=1)
        q.put((tuple(full.shape), bool(torch.equal(full, ref))))
    dist.barrier()
    dist.destroy_process_group()


def _run(T):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, T, q)) for r in range(2)]
    for p in procs:
        p.start()
    shape, same = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return shape, same


def test_two_rank_frame_sharding_even():
    shape, same = _run(6)
    assert shape == (6 * 4, 144) and same


def test_two_rank_frame_sharding_ragged():
    shape, same = _run(5)
    assert shape == (5 * 4, 144) and same


def _seg_mean(emb, lens):
    """stand-in for Qwen2Engine.forward_segments on CPU (per-sequence, no interaction between sequences)"""
    out, b = [], 0
    for n in lens:
        out.append(torch.tanh(emb[b:b + n].float()).mean(dim=0))
        b += n
    return torch.stack(out)


def _events_worker(rank, world, port, lens, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from cogstream_amd.parallel import pooled_means_sharded
    g = torch.Generator().manual_seed(5)
    segs = [torch.randn(n, 32, generator=g) for n in lens]
    calls = []

    def fwd(emb, ls):
        calls.append(list(ls))
        return _seg_mean(emb, ls)

    got = pooled_means_sharded(fwd, segs, rank, world)
    ref = _seg_mean(torch.cat(segs), lens)
    q.put((rank, bool(torch.equal(got, ref)), calls))
    dist.barrier()
    dist.destroy_process_group()


def test_event_summary_sequences_shard_over_two_ranks():
    from cogstream_amd.parallel import partition_sequences
    lens = [40, 7, 33, 12, 25, 5, 19]          # K = 6 event prompts + the question
    plan = partition_sequences(lens, 2)
    assert sorted(plan[0] + plan[1]) == list(range(7))
    assert abs(sum(lens[i] for i in plan[0]) - sum(lens[i] for i in plan[1])) <= max(lens) // 4
    assert partition_sequences([3, 2], 4) == [[0], [1], [], []]      # more ranks than sequences
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_events_worker, args=(r, 2, port, lens, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, same, calls in res:
        assert same
        assert calls == [[lens[i] for i in plan[rank]]]     # ONE var-len forward per rank, over its share only


def test_frame_plan_cuts_at_frame_boundaries_and_covers_every_frame():
    """parallel.FramePlan: contiguous runs of whole frames, balanced by patches, every frame exactly once, the same
    plan on every rank; more ranks than frames leaves ranks empty"""
    from cogstream_amd.parallel import FramePlan
    p = FramePlan(torch.tensor([[256, 10, 20]]), torch.tensor([2]), 8)        # BASELINE configs[2] over 8 GPUs
    assert p.pieces == [[(0, 32 * r, 32 * r + 32)] for r in range(8)] and p.token_counts == [1600] * 8
    assert p.patch_rows[3] == (19200, 25600)
    grids, merges = torch.tensor([[3, 4, 6], [2, 2, 4], [5, 4, 4]]), torch.tensor([2, 2, 2])
    for world in (1, 2, 3, 4, 7, 16):
        p = FramePlan(grids, merges, world)
        flat = [(v, f) for pcs in p.pieces for v, b, e in pcs for f in range(b, e)]
        assert flat == [(v, f) for v, t in enumerate((3, 2, 5)) for f in range(t)]
        assert p.patch_rows[0][0] == 0 and p.patch_rows[-1][1] == 3 * 24 + 2 * 8 + 5 * 16
        assert all(a[1] == b[0] for a, b in zip(p.patch_rows, p.patch_rows[1:]))
        assert sum(p.token_counts) == (3 * 24 + 2 * 8 + 5 * 16) // 4
    px, g, m = FramePlan(grids, merges, 4).local(torch.arange(168 * 2).view(168, 2), 1)
    assert g.tolist() == [[1, 4, 6], [1, 2, 4]] and m.tolist() == [2, 2] and px[0, 0].item() == 48 * 2 and px.shape[0] == 32


def _ragged_worker(rank, world, port, counts, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cogstream_amd.parallel import gather_rows
    local = torch.full((counts[rank], 3), float(rank + 1)) + torch.arange(counts[rank])[:, None]
    got = gather_rows(local, counts)
    q.put((rank, got.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_rows_ragged_and_empty_shards():
    ctx = mp.get_context("spawn")
    for counts in ([3, 1], [0, 2], [2, 2]):
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_ragged_worker, args=(r, 2, port, counts, q)) for r in range(2)]
        for p in procs:
            p.start()
        res = dict(q.get(timeout=120) for _ in range(2))
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        want = [[float(r + 1 + i)] * 3 for r, c in enumerate(counts) for i in range(c)]
        assert res[0] == want and res[1] == want


def test_sampler_indices_equal_torch_distributed_sampler():
    """answer_generate.sampler_indices is DistributedSampler(dataset, num_replicas, rank) of the reference driver
    (evaluate/answer_generate.py:186): shuffled under seed 0, padded by wrap-around"""
    from torch.utils.data.distributed import DistributedSampler
    from cogstream_amd.answer_generate import sampler_indices
    for n, w in ((10, 4), (3, 8), (8, 8), (1, 2), (17, 3)):
        for r in range(w):
            assert list(DistributedSampler(list(range(n)), num_replicas=w, rank=r)) == sampler_indices(n, r, w)
            assert list(DistributedSampler(list(range(n)), num_replicas=w, rank=r, shuffle=False)) == sampler_indices(n, r, w, shuffle=False)
