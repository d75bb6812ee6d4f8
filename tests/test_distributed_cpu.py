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
