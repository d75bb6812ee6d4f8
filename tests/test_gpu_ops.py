"""Operator-level parity: every HIP kernel, called through the C ABI, against a plain torch fp32
restatement of the same op on the same seeded inputs. Tolerances are stated per test:
bf16 storage => ~2^-8 relative per rounding; fp32 parity mode => 1e-4 or tighter."""
import math

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu


def _ops():
    from cogstream_amd import ops
    return ops


def _L():
    from cogstream_amd import _lib
    return _lib


@pytest.mark.parametrize("dtype,tol", [(torch.bfloat16, 1.2e-2), (torch.float32, 2e-5)])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 256, 192), (1, 512, 256), (77, 132, 128), (1024, 1152, 1152),
                                   (515, 132, 64), (777, 384, 704)])
def test_gemm_bias_residual(dev, dtype, tol, M, N, K):
    ops = _ops()
    torch.manual_seed(M * 7 + N)
    a = torch.randn(M, K).to(dtype)
    w = (torch.randn(N, K) / math.sqrt(K)).to(dtype)
    b = torch.randn(N).to(dtype)
    r = torch.randn(M, N).to(dtype)
    ref = a.float() @ w.float().t() + b.float() + r.float()
    out = ops.gemm(a.to(dev), w.to(dev), b.to(dev), residual=r.to(dev))
    assert out.dtype == dtype
    assert rel_err(out.float(), ref) < tol


def test_gemm_integer_exact(dev):
    """MFMA fragment/layout check with exactly representable data and an asymmetric W"""
    run_gemm_integer_exact(dev, 256, 256, 128)
    run_gemm_integer_exact(dev, 1024 + 40, 384, 448)   # 256x128 ring-buffered variant, ragged M


def run_gemm_integer_exact(dev, M, N, K):
    ops = _ops()
    a = torch.randint(-3, 4, (M, K)).float()
    w = (torch.arange(N)[:, None] % 5 - 2).float() * torch.randint(0, 2, (N, K)).float() + (torch.arange(K)[None, :] % 3).float()
    ref = a @ w.t()
    out = ops.gemm(a.to(dev).bfloat16(), w.to(dev).bfloat16(), out_f32=True)
    assert torch.equal(out.cpu(), ref)
    out32 = ops.gemm(a.to(dev), w.to(dev))
    assert torch.equal(out32.cpu(), ref)


@pytest.mark.parametrize("act", ["tanh", "erf"])
def test_gemm_gelu(dev, act):
    ops, L = _ops(), _L()
    torch.manual_seed(1)
    a = torch.randn(200, 128).bfloat16()
    w = (torch.randn(256, 128) / 11).bfloat16()
    b = torch.randn(256).bfloat16()
    pre = a.float() @ w.float().t() + b.float()
    ref = F.gelu(pre, approximate="tanh") if act == "tanh" else F.gelu(pre)
    out = ops.gemm(a.to(dev), w.to(dev), b.to(dev), act=L.ACT_GELU_TANH if act == "tanh" else L.ACT_GELU_ERF)
    assert rel_err(out.float(), ref) < 1.2e-2


@pytest.mark.parametrize("M", [1, 130])
def test_gemm_swiglu(dev, M):
    ops, L = _ops(), _L()
    torch.manual_seed(2)
    H, I = 128, 192
    x = torch.randn(M, H).bfloat16()
    wg = (torch.randn(I, H) / 11).bfloat16()
    wu = (torch.randn(I, H) / 11).bfloat16()
    ref = F.silu(x.float() @ wg.float().t()) * (x.float() @ wu.float().t())
    gu = torch.stack([wg, wu], dim=1).reshape(2 * I, H).contiguous()
    out = ops.gemm(x.to(dev), gu.to(dev), act=L.ACT_SWIGLU)
    assert out.shape == (M, I)
    assert rel_err(out.float(), ref) < 1.2e-2


@pytest.mark.parametrize("M,heads,interleaved", [(1, 2, False), (96, 2, False), (96, 2, True), (1300, 2, True),
                                                 (1300, 7, True), (1300, 7, False)])
def test_gemm_rope_epilogue(dev, M, heads, interleaved):
    """columns [0, rope_cols) rotated as rotate_half pairs after the (d, d+hd/2) row interleave; both table
    formats; M = 1300 reaches the 256x128 (heads 2) and ping-pong (heads 7) kernels, whose interior wave tiles
    take the lean epilogue while the tile straddling rope_cols and the ragged last row block take the general one"""
    ops = _ops()
    torch.manual_seed(3)
    hd, K = 72, 128
    N = 3 * heads * hd  # q | k | v, v not rotated
    x = torch.randn(M, K).bfloat16()
    w = (torch.randn(N, K) / 11).bfloat16()
    b = torch.randn(N).bfloat16()
    ang = torch.rand(M, hd // 2) * 6.0
    y = (x.float() @ w.float().t() + b.float()).view(M, 3 * heads, hd)
    cos = torch.cat([ang.cos(), ang.cos()], -1)[:, None, :]
    sin = torch.cat([ang.sin(), ang.sin()], -1)[:, None, :]
    rot = torch.cat([-y[..., hd // 2:], y[..., :hd // 2]], -1)
    ref = y.clone()
    ref[:, :2 * heads] = (y * cos + rot * sin)[:, :2 * heads]
    # pack: within each q/k head, new row 2i <- d=i, 2i+1 <- d=i+hd/2
    perm = torch.arange(hd).view(2, hd // 2).t().reshape(-1)
    wp = w.view(3 * heads, hd, K).clone()
    bp = b.view(3 * heads, hd).clone()
    wp[:2 * heads] = wp[:2 * heads][:, perm]
    bp[:2 * heads] = bp[:2 * heads][:, perm]
    out = ops.gemm(x.to(dev), wp.reshape(N, K).contiguous().to(dev), bp.reshape(N).contiguous().to(dev),
                   rope_cos=(torch.stack([ang.cos(), ang.sin()], -1) if interleaved else ang.cos()).contiguous().to(dev),
                   rope_sin=None if interleaved else ang.sin().contiguous().to(dev),
                   rope_cols=2 * heads * hd, head_dim=hd)
    refp = ref.clone()
    refp[:, :2 * heads] = ref[:, :2 * heads][:, :, perm]
    assert rel_err(out.float().view(M, 3 * heads, hd), refp) < 1.2e-2


def test_gemm_rope_position_lut_equals_row_table(dev):
    """the ping-pong GEMM's LDS position LUT (2-D rotary: first half of a head's pairs uses the row's h, second half
    its w) must give bit-identical outputs to the per-row (cos, sin) table built from the same angles -- and both
    must match the plain reference"""
    ops = _ops()
    torch.manual_seed(17)
    hd, heads, K, M = 72, 7, 128, 1500            # N = 1512: ping-pong kernel; ragged last row block; rope straddle
    nf, maxpos = hd // 4, 45
    N = 3 * heads * hd
    x = torch.randn(M, K).bfloat16()
    w = (torch.randn(N, K) / 11).bfloat16()
    b = torch.randn(N).bfloat16()
    hpos, wpos = torch.randint(0, 23, (M,)), torch.randint(0, maxpos, (M,))
    inv_freq = 1.0 / (10000.0 ** (torch.arange(nf, dtype=torch.float32) / nf))
    lut_ang = torch.arange(maxpos, dtype=torch.float32)[:, None] * inv_freq[None, :]            # [maxpos, nf]
    lut = torch.stack([lut_ang.cos(), lut_ang.sin()], -1).contiguous()                          # [maxpos, nf, 2]
    ang = torch.cat([hpos[:, None].float() * inv_freq[None, :], wpos[:, None].float() * inv_freq[None, :]], 1)   # [M, hd/2]
    table = torch.cat([lut[hpos], lut[wpos]], 1).contiguous()                                   # [M, hd/2, 2] same values
    assert torch.equal(table[..., 0], torch.cat([lut_ang[hpos], lut_ang[wpos]], 1).cos())
    rowpos = (hpos | (wpos << 16)).to(torch.int32)
    perm = torch.arange(hd).view(2, hd // 2).t().reshape(-1)
    wp = w.view(3 * heads, hd, K).clone()
    bp = b.view(3 * heads, hd).clone()
    wp[:2 * heads] = wp[:2 * heads][:, perm]
    bp[:2 * heads] = bp[:2 * heads][:, perm]
    args = dict(rope_cos=table.to(dev), rope_sin=None, rope_cols=2 * heads * hd, head_dim=hd)
    xa, wa, ba = x.to(dev), wp.reshape(N, K).contiguous().to(dev), bp.reshape(N).contiguous().to(dev)
    out_tab = ops.gemm(xa, wa, ba, **args)
    out_lut = ops.gemm(xa, wa, ba, rope_lut=lut.to(dev), rope_rowpos=rowpos.to(dev), **args)
    assert torch.equal(out_lut, out_tab)
    y = (x.float() @ w.float().t() + b.float()).view(M, 3 * heads, hd)
    cos = torch.cat([ang.cos(), ang.cos()], -1)[:, None, :]
    sin = torch.cat([ang.sin(), ang.sin()], -1)[:, None, :]
    rot = torch.cat([-y[..., hd // 2:], y[..., :hd // 2]], -1)
    ref = y.clone()
    ref[:, :2 * heads] = (y * cos + rot * sin)[:, :2 * heads]
    ref[:, :2 * heads] = ref[:, :2 * heads][:, :, perm]
    assert rel_err(out_lut.float().view(M, 3 * heads, hd), ref) < 1.2e-2


def _attn_ref(q, k, v, hq, hkv, hd, cu=None, causal=False, q_pos0=0, row_lo=None, row_hi=None, bias=0.0, scale=None):
    Lq, Lk = q.shape[0], k.shape[0]
    q = q.float().view(Lq, hq, hd).transpose(0, 1)
    k = k.float().view(Lk, hkv, hd).transpose(0, 1).repeat_interleave(hq // hkv, 0)
    v = v.float().view(Lk, hkv, hd).transpose(0, 1).repeat_interleave(hq // hkv, 0)
    s = q @ k.transpose(1, 2) * (1.0 / math.sqrt(hd) if scale is None else scale)
    qi, kj = torch.arange(Lq)[:, None], torch.arange(Lk)[None, :]
    allow = torch.ones(Lq, Lk, dtype=torch.bool)
    if cu is not None:
        seg_q = torch.bucketize(qi, cu[1:], right=True)
        seg_k = torch.bucketize(kj, cu[1:], right=True)
        allow &= seg_q == seg_k
    if causal:
        allow &= kj <= qi + q_pos0
    if row_lo is not None:
        s = s + ((kj >= row_lo[:, None]) & (kj < row_hi[:, None])).float() * bias
    s = s.masked_fill(~allow, float("-inf"))
    return (F.softmax(s, -1) @ v).transpose(0, 1).reshape(Lq, hq * hd)


@pytest.mark.parametrize("hd,heads,lens", [(72, 2, [256, 256]), (72, 3, [200, 200, 200]), (72, 2, [924]), (72, 1, [37, 130])])
def test_attention_block_diag_bf16(dev, hd, heads, lens):
    ops = _ops()
    torch.manual_seed(sum(lens))
    n = sum(lens)
    qkv = torch.randn(n, 3 * heads * hd).bfloat16()
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32)
    H = heads * hd
    ref = _attn_ref(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], heads, heads, hd, cu=cu.long())
    g = qkv.to(dev)
    out = ops.attention(g[:, :H], g[:, H:2 * H], g[:, 2 * H:], hq=heads, hkv=heads, head_dim=hd,
                        cu_seqlens=cu.to(dev), max_seqlen=max(lens))
    assert rel_err(out.float(), ref) < 1.5e-2


LOG2E = 1.4426950408889634


@pytest.mark.parametrize("hd,heads,lens", [(72, 2, [256, 256]), (72, 2, [924]), (72, 1, [37, 130]), (128, 2, [300, 64])])
def test_attention_prescaled_q_block_diag(dev, hd, heads, lens):
    """q_prescaled: Q carries scale*log2(e) (rounded to bf16 once, as the QKV GEMM epilogue does) and the kernel
    runs the deferred-max softmax; the reference is the plain softmax of the same rounded Q with scale ln 2"""
    ops = _ops()
    torch.manual_seed(sum(lens) + hd)
    n = sum(lens)
    H = heads * hd
    qkv = torch.randn(n, 3 * H)
    qkv[:, :H] *= LOG2E / math.sqrt(hd)
    # rows whose scores move by tens of log2 units from tile to tile exercise the rebase path
    qkv[5, :H] *= 6.0
    qkv = qkv.bfloat16()
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32)
    ref = _attn_ref(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], heads, heads, hd, cu=cu.long(), scale=math.log(2.0))
    g = qkv.to(dev)
    out = ops.attention(g[:, :H], g[:, H:2 * H], g[:, 2 * H:], hq=heads, hkv=heads, head_dim=hd,
                        cu_seqlens=cu.to(dev), max_seqlen=max(lens), q_prescaled=True)
    assert rel_err(out.float(), ref) < 1.5e-2


@pytest.mark.parametrize("heads,lens", [
    (1, [1]), (1, [31]), (2, [32, 33]), (1, [63, 64, 65]), (2, [96, 127, 128, 129]), (1, [160, 191, 192, 193]),
    (8, [200] * 5), (2, [255, 256, 257]), (1, [320, 384, 449]), (8, [924, 924]), (3, [1024, 70, 1000]), (16, [130] * 4)])
@pytest.mark.parametrize("data", ["random", "ramp"])
def test_attention_vit_pipeline_edges(dev, heads, lens, data):
    """The encoder's production attention kernel (csrc/attn_vit.hip: hd 72, per-frame segments, pre-scaled Q,
    software-pipelined at 32-key blocks with a 4-slot LDS-DMA ring) over every seam of its control flow: segments
    shorter than a block / a tile / the three-tile prologue, ragged and exactly-full last blocks and tiles, ragged and
    empty query blocks (waves that only stage), both workgroup orders (frames x heads a multiple of 8 or not).
    "ramp": the scores of every row GROW along the keys by ~2 log2 units per key block, so the deferred maximum is
    moved (O rescaled, the block's scores recomputed) again and again in steady state, and one row sits far below
    zero. Checked row by row against the fp32 softmax of the same bf16 inputs."""
    ops = _ops()
    hd, H = 72, heads * 72
    n = sum(lens)
    g = torch.Generator().manual_seed(1000 * heads + n + (data == "ramp"))
    qkv = torch.randn(n, 3 * H, generator=g)
    qkv[:, :H] *= LOG2E / math.sqrt(hd)
    if data == "ramp":
        # q = a common direction u (scaled), k = u * (position in the segment / 16): score ~ 1.5 * pos / 16 log2 units
        u = torch.randn(hd, generator=g)
        u /= u.norm()
        pos = torch.cat([torch.arange(L, dtype=torch.float32) for L in lens])
        for hh in range(heads):
            qkv[:, hh * hd:(hh + 1) * hd] += 1.5 * u
            qkv[:, H + hh * hd:H + (hh + 1) * hd] += u * (pos[:, None] / 16.0)
        qkv[n // 2, :H] = -3.0 * u.repeat(heads)          # a row whose scores fall with the key index
    qkv = qkv.bfloat16()
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32)
    ref = _attn_ref(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], heads, heads, hd, cu=cu.long(), scale=math.log(2.0))
    gq = qkv.to(dev)
    out = ops.attention(gq[:, :H], gq[:, H:2 * H], gq[:, 2 * H:], hq=heads, hkv=heads, head_dim=hd,
                        cu_seqlens=cu.to(dev), max_seqlen=max(lens), q_prescaled=True).float().cpu()
    assert torch.isfinite(out).all()
    # per row: error against the row's own largest output (a dropped or doubled block shows up at 1e-1 .. 1)
    err = (out - ref).abs().view(n, heads, hd).amax(2) / ref.abs().view(n, heads, hd).amax(2).clamp_min(1e-3)
    assert float(err.max()) < 3e-2, (float(err.max()), int(err.argmax()) // heads)
    # convexity: every output lies inside the range its segment's V spans
    v = qkv[:, 2 * H:].float()
    for i, L in enumerate(lens):
        a, b = int(cu[i]), int(cu[i + 1])
        assert (out[a:b] <= v[a:b].amax(0) + 2e-2).all() and (out[a:b] >= v[a:b].amin(0) - 2e-2).all()


@pytest.mark.parametrize("seg", [129, 132, 160, 161, 176, 192, 193, 200, 240, 256, 257, 264, 308, 320, 336, 360, 384, 396, 440, 448,
                                 500, 512, 576, 924, 1024])
@pytest.mark.parametrize("layout", ["token", "head"])
def test_attention_vit_compile_time_end_equals_run_time_end_bit_for_bit(dev, seg, layout):
    """One video = every segment the same length: the pipelined ViT kernel is launched as the instantiation whose ragged end
    (tiles behind the four-tile loop R = 4..7, 32-key blocks of the last tile 1 / 2) has that length's shape as template
    parameters (csrc/attn_vit.hip, round 6). Lengths for all eight (R, blocks) pairs, last blocks of 1, 8, 31, 32 keys, exactly
    full last tiles, the loop running 0 / 1 / 3 trips, both K/V layouts, growing scores on half the heads. The launch with the
    hint must equal the launch without it (run-time end) bit for bit, and both the fp32 softmax."""
    ops = _ops()
    from cogstream_amd import _lib as L
    heads, nseg, hd = 4, 5, 72
    H, n = heads * hd, nseg * seg
    g = torch.Generator().manual_seed(seg)
    qkv = torch.randn(n, 3 * H, generator=g)
    qkv[:, :H] *= LOG2E / math.sqrt(hd)
    u = torch.randn(hd, generator=g)
    u /= u.norm()
    pos = torch.arange(seg, dtype=torch.float32).repeat(nseg)
    for hh in range(0, heads, 2):
        qkv[:, hh * hd:(hh + 1) * hd] += 1.5 * u
        qkv[:, H + hh * hd:H + (hh + 1) * hd] += u * (pos[:, None] / 16.0)
    qkv = qkv.bfloat16()
    cu = torch.arange(nseg + 1, dtype=torch.int32) * seg
    ref = _attn_ref(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], heads, heads, hd, cu=cu.long(), scale=math.log(2.0))
    if layout == "head":
        hm = qkv.view(n, 3, heads, hd).permute(1, 2, 0, 3).contiguous().to(dev)        # [which][head][row][hd]
        run = lambda: ops.attention(hm[0], hm[1], hm[2], hq=heads, hkv=heads, head_dim=hd, cu_seqlens=cu.to(dev),
                                    max_seqlen=seg, q_prescaled=True, head_major=True)
    else:
        gq = qkv.to(dev)
        run = lambda: ops.attention(gq[:, :H], gq[:, H:2 * H], gq[:, 2 * H:], hq=heads, hkv=heads, head_dim=hd,
                                    cu_seqlens=cu.to(dev), max_seqlen=seg, q_prescaled=True)
    plain = run()
    assert L.debug_get("attn_vit_last_end") == 0
    with L.debug_switch("attn_uniform_hint", seg):
        hinted = run()
        nt = (seg + 63) // 64
        want = 10 * (4 + ((nt - 4) & 3)) + (2 if seg - 64 * (nt - 1) > 32 else 1) if nt >= 4 else 0
        assert L.debug_get("attn_vit_last_end") == want
        with L.debug_switch("attn_vit_len", 0):
            hinted_rt = run()
            assert L.debug_get("attn_vit_last_end") == 0
    assert torch.equal(plain, hinted) and torch.equal(plain, hinted_rt)
    out = hinted.float().cpu()
    err = (out - ref).abs().view(n, heads, hd).amax(2) / ref.abs().view(n, heads, hd).amax(2).clamp_min(1e-3)
    assert float(err.max()) < 3e-2, (float(err.max()), int(err.argmax()) // heads)


@pytest.mark.parametrize("S,pos0", [(200, 0), (130, 77), (1, 300), (1, 5000)])
def test_attention_prescaled_q_causal_gqa(dev, S, pos0):
    ops = _ops()
    torch.manual_seed(S + 1)
    hd, hq, hkv = 128, 4, 2
    ctx = pos0 + S
    q = (torch.randn(S, hq * hd) * (LOG2E / math.sqrt(hd))).bfloat16()
    k = torch.randn(ctx, hkv * hd).bfloat16()
    v = torch.randn(ctx, hkv * hd).bfloat16()
    ref = _attn_ref(q, k, v, hq, hkv, hd, causal=True, q_pos0=pos0, scale=math.log(2.0))
    out = ops.attention(q.to(dev), k.to(dev), v.to(dev), hq=hq, hkv=hkv, head_dim=hd, causal=True, q_pos0=pos0,
                        q_prescaled=True)
    assert rel_err(out.float(), ref) < 1.5e-2
    if S == 1:
        out2 = ops.attention(q.to(dev), k.to(dev), v.to(dev), hq=hq, hkv=hkv, head_dim=hd, causal=True, q_pos0=pos0,
                             nsplit=3 if pos0 < 1000 else 20, q_prescaled=True)
        assert rel_err(out2.float(), ref) < 1.5e-2


@pytest.mark.parametrize("hq,hkv", [(28, 4), (16, 1), (4, 4), (6, 2)])
@pytest.mark.parametrize("ctx,nsplit", [(1, 2), (63, 2), (64, 3), (65, 2), (255, 4), (256, 4), (257, 5), (1000, 8), (1000, 33),
                                        (4100, 17)])
def test_decode_attention_kernel_edges(dev, hq, hkv, ctx, nsplit):
    """The generated tokens' attention (csrc/attn_decode.hip: one query row, every wave owns whole 64-key tiles, four
    waves merged in LDS, partials combined by attn_combine_kernel): contexts around every tile boundary, splits with a
    ragged last tile, with fewer tiles than waves and with no tile at all (nsplit > tiles), 1 / 2 / 7 / 16 query heads
    per key/value head. Checked per head against the fp32 softmax of the same bf16 inputs."""
    ops = _ops()
    hd = 128
    g = torch.Generator().manual_seed(ctx * 131 + nsplit * 7 + hq)
    q = (torch.randn(1, hq * hd, generator=g) * (LOG2E / math.sqrt(hd))).bfloat16()
    k = torch.randn(ctx, hkv * hd, generator=g).bfloat16()
    v = torch.randn(ctx, hkv * hd, generator=g).bfloat16()
    if ctx > 4:
        k[ctx // 2] *= 4.0                       # one dominant key in the middle: the running maximum moves inside a split
    ref = _attn_ref(q, k, v, hq, hkv, hd, causal=True, q_pos0=ctx - 1, scale=math.log(2.0))
    out = ops.attention(q.to(dev), k.to(dev), v.to(dev), hq=hq, hkv=hkv, head_dim=hd, causal=True, q_pos0=ctx - 1,
                        nsplit=nsplit, q_prescaled=True).float().cpu()
    assert torch.isfinite(out).all()
    err = (out - ref).abs().view(hq, hd).amax(1) / ref.abs().view(hq, hd).amax(1).clamp_min(1e-3)
    assert float(err.max()) < 2e-2, (float(err.max()), int(err.argmax()))


def test_attention_integer_exact(dev):
    """layout check: one-hot attention (huge scale) must copy integer V rows exactly"""
    ops = _ops()
    hd, n = 128, 192
    torch.manual_seed(0)
    tgt = torch.randperm(n)
    # unique two-hot keys: q = 256*k[tgt] scores 512 on its target and <= 256 elsewhere (others underflow to 0)
    k = torch.zeros(n, hd)
    k[torch.arange(n), torch.arange(n) % 120] = 1.0
    k[torch.arange(n), 120 + torch.arange(n) // 120] = 1.0
    q = k[tgt] * 256.0
    v = torch.randint(-8, 9, (n, hd)).float()
    out = ops.attention(q.bfloat16().to(dev), k.bfloat16().to(dev), v.bfloat16().to(dev), hq=1, hkv=1, head_dim=hd,
                        scale=1.0)
    assert torch.equal(out.float().cpu(), v[tgt])


@pytest.mark.parametrize("S,pos0", [(200, 0), (130, 77), (1, 300)])
def test_attention_causal_gqa_bf16(dev, S, pos0):
    ops = _ops()
    torch.manual_seed(S)
    hd, hq, hkv = 128, 4, 2
    ctx = pos0 + S
    q = torch.randn(S, hq * hd).bfloat16()
    k = torch.randn(ctx, hkv * hd).bfloat16()
    v = torch.randn(ctx, hkv * hd).bfloat16()
    ref = _attn_ref(q, k, v, hq, hkv, hd, causal=True, q_pos0=pos0)
    out = ops.attention(q.to(dev), k.to(dev), v.to(dev), hq=hq, hkv=hkv, head_dim=hd, causal=True, q_pos0=pos0)
    assert rel_err(out.float(), ref) < 1.5e-2
    if S == 1:
        out2 = ops.attention(q.to(dev), k.to(dev), v.to(dev), hq=hq, hkv=hkv, head_dim=hd, causal=True, q_pos0=pos0,
                             nsplit=3)
        assert rel_err(out2.float(), ref) < 1.5e-2


@pytest.mark.parametrize("dtype,tol,rowwise", [(torch.bfloat16, 1.5e-2, False), (torch.float32, 1e-5, True)])
def test_attention_eager_global_bias(dev, dtype, tol, rowwise):
    ops = _ops()
    torch.manual_seed(5)
    hd, heads, lens = 72, 2, [64, 64, 64]
    n = sum(lens)
    H = heads * hd
    qkv = torch.randn(n, 3 * H).to(dtype)
    lo = torch.tensor([64 * (i // 64) for i in range(n)], dtype=torch.int32)
    hi = lo + 64
    ref = _attn_ref(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], heads, heads, hd, row_lo=lo.long(), row_hi=hi.long(), bias=1.0)
    g = qkv.to(dev)
    out = ops.attention(g[:, :H], g[:, H:2 * H], g[:, 2 * H:], hq=heads, hkv=heads, head_dim=hd, row_lo=lo.to(dev),
                        row_hi=hi.to(dev), bias=1.0, force_rowwise=rowwise)
    assert rel_err(out.float(), ref) < tol


@pytest.mark.parametrize("dtype,tol", [(torch.bfloat16, 1e-2), (torch.float32, 1e-5)])
@pytest.mark.parametrize("H", [144, 1152, 3584])
def test_norms(dev, dtype, tol, H):
    ops = _ops()
    torch.manual_seed(H)
    x = (torch.randn(37, H) * 2 + 0.3).to(dtype)
    g = (1 + 0.1 * torch.randn(H)).to(dtype)
    b = (0.1 * torch.randn(H)).to(dtype)
    ref = F.layer_norm(x.float(), (H,), g.float(), b.float(), 1e-6)
    assert rel_err(ops.layernorm(x.to(dev), g.to(dev), b.to(dev)).float(), ref) < tol
    xf = x.float()
    ref = g.float() * (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6))
    assert rel_err(ops.rmsnorm(x.to(dev), g.to(dev)).float(), ref) < tol
    xm = x[:36]
    ref = F.layer_norm(xm.float(), (H,), g.float(), b.float(), 1e-6).view(9, 4, H).mean(1)
    assert rel_err(ops.ln_merge(xm.contiguous().to(dev), g.to(dev), b.to(dev), 4).float(), ref) < tol


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_pixdiff_mask_matches_torch_semantics(dev, dtype):
    """bit-exact against the torch expression of cogreasoner_chat.py:407-414 run in the same dtype"""
    ops = _ops()
    torch.manual_seed(11)
    t, P, E = 6, 20, 2352
    base = torch.rand(1, P, E) * 2 - 1
    pix = base.repeat(t, 1, 1)
    pix[1] += 0.002 * torch.randn(P, E)           # tiny change: mostly below threshold
    pix[2, :5] += 0.5 * torch.randn(5, E)         # big change on 5 tokens
    pix[4] = pix[3]                               # identical frame -> min_tokens kicks in
    pix = pix.to(dtype)
    diff = torch.abs(pix[1:] - pix[:-1]).mean(dim=-1) * 255
    diff = torch.cat([torch.full_like(diff[0:1], 1.1), diff], dim=0)
    m = diff > 0.1
    pad = torch.nonzero(m.sum(dim=1) < 1)[:, 0]
    m[pad, :1] = 1
    minor = torch.zeros(t, dtype=torch.uint8)
    minor[5] = 1
    m[5, 0] = True
    m[5, 1:] = False
    out = ops.pixdiff_mask(pix.reshape(-1, 588).contiguous().to(dev), t, P, minor=minor.to(dev))
    assert torch.equal(out.cpu().bool(), m.flatten())
    assert 0 < int(m.sum()) < m.numel()


def test_frame_mean_gather_mean_cosine(dev):
    ops = _ops()
    torch.manual_seed(12)
    P, D, T = 7, 256, 5
    f = torch.randn(T * P, D).bfloat16()
    ref = f.view(T, P, D).clone()
    for i in (1, 3):
        ref[i, 0] = ref[i].mean(dim=0)
    g = f.to(dev).clone()
    ops.frame_mean_to_slot0(g, P, torch.tensor([1, 3], dtype=torch.int32, device=dev))
    assert rel_err(g.float().cpu(), ref.view(-1, D).float()) < 8e-3
    ta, tb = torch.randn(50, D).bfloat16(), torch.randn(9, D).bfloat16()
    idx = torch.tensor([3, -1, 49, -9, 0, -4], dtype=torch.int64)
    out = ops.gather_rows(ta.to(dev), tb.to(dev), idx.to(dev))
    ref = torch.stack([ta[i] if i >= 0 else tb[-i - 1] for i in idx.tolist()])
    assert torch.equal(out.cpu(), ref)
    x = torch.randn(333, D).bfloat16()
    assert rel_err(ops.mean_rows(x.to(dev)), x.float().mean(0)) < 1e-5
    a, b = torch.randn(D), torch.randn(6, D)
    assert rel_err(ops.cosine(a.to(dev), b.to(dev)), F.cosine_similarity(a[None], b, dim=1)) < 1e-5


def test_logits_ops(dev):
    ops = _ops()
    torch.manual_seed(13)
    n = 152064
    lg = torch.randn(n)
    lg[777] = lg.max() + 1
    assert int(ops.argmax(lg.to(dev))) == 777
    prev = torch.tensor([5, 777, 5, 42], dtype=torch.int64)
    ref = lg.clone()
    g = ref[prev]
    ref[prev] = torch.where(g < 0, g * 1.05, g / 1.05)
    ref = ref / 0.7
    x = lg.to(dev).clone()
    ops.logits_process(x, prev.to(dev), 1.05, None, 0.7)
    assert torch.allclose(x.cpu(), ref, rtol=1e-6, atol=0)
    allowed = torch.tensor([11, 15, 16, 58, 60, 151645], dtype=torch.int32)
    x = lg.to(dev).clone()
    ops.logits_process(x, None, 1.0, allowed.to(dev), 1.0)
    refm = torch.full_like(lg, float("-inf"))
    refm[allowed.long()] = lg[allowed.long()]
    assert torch.equal(x.cpu(), refm)
    val, idx = ops.topk(lg.to(dev), 20)
    rv, ri = torch.topk(lg, 20)
    assert torch.equal(val.cpu(), rv) and torch.equal(idx.cpu().long(), ri)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_kmeans_steps(dev, dtype):
    ops = _ops()
    torch.manual_seed(14)
    T, PD, K = 40, 2 * 1024 + 512 + 8, 5
    cent = torch.randn(K, PD) * 3
    lab = torch.randint(0, K, (T,))
    x = (cent[lab] + 0.3 * torch.randn(T, PD)).to(dtype)
    ts = torch.arange(T).float()
    ws = ops.kmeans_workspace(T, PD, K, dev)
    rows = torch.tensor([0, 7, 13, 21, 30], dtype=torch.int32)
    xg = x.to(dev)
    d2 = ops.kmeans_sqdist(xg, None, rows.to(dev), K, ws)
    ref = torch.cdist(x.double(), x[rows.long()].double()) ** 2
    assert rel_err(d2, ref) < 1e-5
    cf = x[rows.long()].float().contiguous()
    ct = ts[rows.long()].contiguous()
    d2b = ops.kmeans_sqdist(xg, cf.to(dev), None, K, ws)
    assert rel_err(d2b, ref) < 1e-5
    assign, counts = ops.kmeans_assign(d2b, ts.to(dev), ct.to(dev), 2.0)
    df = ref.float().sqrt()
    dtm = (ts[:, None] - ct[None]).abs()
    nf = (df - df.min(1, keepdim=True).values) / (df.max(1, keepdim=True).values - df.min(1, keepdim=True).values)
    nt = (dtm - dtm.min(1, keepdim=True).values) / (dtm.max(1, keepdim=True).values - dtm.min(1, keepdim=True).values)
    ra = torch.sqrt(nf ** 2 + 2 * nt ** 2).argmin(1)
    assert torch.equal(assign.cpu(), ra)
    assert torch.equal(counts.cpu().long(), torch.bincount(ra, minlength=K))
    # force an empty cluster to exercise the reseed path
    a2 = ra.clone()
    a2[a2 == 4] = 0
    reseed = torch.tensor([0, 0, 0, 0, 17], dtype=torch.int32)
    cg, ctg = cf.to(dev).clone(), ct.to(dev).clone()
    shift = ops.kmeans_update(xg, ts.to(dev), a2.to(dev), reseed.to(dev), cg, ctg, ws)
    ncf, nct = torch.zeros_like(cf), torch.zeros_like(ct)
    for i in range(K):
        m = a2 == i
        if m.any():
            ncf[i] = x.float()[m].mean(0)
            nct[i] = ts[m].mean()
        else:
            ncf[i] = x.float()[17]
            nct[i] = ts[17]
    rs = torch.norm(ncf - cf, dim=1).sum() + torch.norm(nct - ct)
    assert rel_err(cg, ncf) < 1e-5 and rel_err(ctg, nct) < 1e-6
    assert abs(float(shift) - float(rs)) / float(rs) < 1e-5


@pytest.mark.parametrize("shape", [(3, 120, 214), (2, 56, 84), (4, 480, 854)])
def test_gpu_preprocessing_bit_exact_vs_pil_path(dev, shape):
    """uint8 frames -> pixel_values on the GPU == the PIL host path (fp32 bit-identical, bf16 = one rounding)"""
    from cogstream_amd import processing as pr
    from cogstream_amd.preprocess_gpu import preprocess_videos_gpu
    t, h, w = shape
    frames, _ = pr.synthetic_clip(t, h, w, kind="drift", clip_idx=3)
    ref = pr.preprocess_videos([frames])
    got = preprocess_videos_gpu([torch.from_numpy(frames).to(dev)], out_dtype=torch.float32)
    assert got["grid_sizes"].tolist() == ref["grid_sizes"].tolist()
    assert torch.equal(got["pixel_values"].cpu(), torch.from_numpy(ref["pixel_values"]))
    got16 = preprocess_videos_gpu([torch.from_numpy(frames).to(dev)], out_dtype=torch.bfloat16)
    assert torch.equal(got16["pixel_values"].cpu(), torch.from_numpy(ref["pixel_values"]).bfloat16())


def test_pair_epilogue_and_relaxed_vmcnt_equal_the_generic_build_bit_for_bit(dev):
    """The ping-pong GEMM's lean pair epilogue lets the next tile's first K-tile waits leave its stores in flight
    (`s_waitcnt vmcnt(8 + ops)`, csrc/gemm.hip). libcogs_hip_nopair.so is the same source built with
    -DCOGS_EPI_NOPAIR: generic epilogue, conservative waits. On the cfg2 ViT shapes (plain, bias, bias+residual,
    bias+GELU, bias+rotary table, bias+rotary LUT) and a Qwen2-prefill shape the two builds must agree bit for bit:
    a wait that lets a wave read a ring slot before its LDS-DMA pieces landed would show up here."""
    import ctypes as C
    from cogstream_amd import _lib as L2
    from cogstream_amd.build import LIB_NOPAIR
    ops = _ops()
    assert LIB_NOPAIR.exists(), "build it with `python -m cogstream_amd.build --nopair` (__graft_entry__.build does)"
    alt = C.CDLL(str(LIB_NOPAIR))
    stamp = LIB_NOPAIR.parent / "csrc" / "build" / "gemm.epi_check.txt"
    assert stamp.read_text().startswith("ok:"), stamp.read_text()      # the build's disassembly check passed
    torch.manual_seed(5)
    M = 59136 // 4                                     # a quarter of cfg2's patches: 58 row blocks, several rounds
    hd, heads = 72, 16
    H, I = 1152, 4352
    g = torch.Generator(device=dev).manual_seed(1)
    rnd = lambda *s: (torch.randn(*s, generator=g, device=dev) * 0.5).bfloat16()
    x, big = rnd(M, H), rnd(M, I)
    nf, maxpos = hd // 4, 42
    hpos = torch.randint(0, 22, (M,), generator=g, device=dev)
    wpos = torch.randint(0, maxpos, (M,), generator=g, device=dev)
    inv_freq = 1.0 / (10000.0 ** (torch.arange(nf, dtype=torch.float32, device=dev) / nf))
    lut_ang = torch.arange(maxpos, dtype=torch.float32, device=dev)[:, None] * inv_freq[None, :]
    lut = torch.stack([lut_ang.cos(), lut_ang.sin()], -1).contiguous()
    rowpos = (hpos | (wpos << 16)).to(torch.int32)
    ang = torch.cat([hpos[:, None].float() * inv_freq, wpos[:, None].float() * inv_freq], 1)
    table = torch.stack([ang.cos(), ang.sin()], -1).contiguous()
    cases = [
        ("qkv+rope table", dict(a=x, w=rnd(3 * H, H), bias=rnd(3 * H), rope_cos=table, rope_cols=2 * H, head_dim=hd)),
        ("qkv+rope lut", dict(a=x, w=rnd(3 * H, H), bias=rnd(3 * H), rope_cos=table, rope_cols=2 * H, head_dim=hd,
                              rope_lut=lut, rope_rowpos=rowpos)),
        ("out-proj+res", dict(a=x, w=rnd(H, H), bias=rnd(H), residual=rnd(M, H))),
        ("fc1+gelu", dict(a=x, w=rnd(I, H), bias=rnd(I), act=L2.ACT_GELU_TANH)),
        ("fc2+res", dict(a=big, w=rnd(H, I), bias=rnd(H), residual=rnd(M, H))),
        ("plain", dict(a=x, w=rnd(3584, H))),
        ("proj gelu-erf", dict(a=x, w=rnd(3584, H), bias=rnd(3584), act=L2.ACT_GELU_ERF)),
        # round 6: the SwiGLU pair epilogue (lane transpose, 16-byte stores) against the generic one's two 2-byte stores
        ("gate/up swiglu", dict(a=x, w=rnd(2 * 2048, H), act=L2.ACT_SWIGLU)),
        ("gate/up swiglu, ragged N", dict(a=x, w=rnd(2 * 2000, H), act=L2.ACT_SWIGLU)),
    ]
    for name, kw in cases:
        for rep in range(2):
            a = ops.gemm(**kw)
            b = ops.gemm(**kw, lib=alt)
            torch.cuda.synchronize()
            assert torch.equal(a, b), (name, rep, float((a.float() - b.float()).abs().max()))


def test_whole_line_gemm_equals_the_k_tile_gemm_bit_for_bit(dev):
    """gemm_tn_pp64_kernel (round 4: 64-wide slabs of whole 128-byte lines in a five-unit ring, three kinds of slab, the
    two wave groups' epilogues in one interval) must produce the bits of gemm_tn_pp_kernel (32-wide K-tiles): same tile,
    same MFMA order, same epilogues. The second kernel is reached through the debug switch gemm_pp64 = 0 (the dispatch
    is asked which body ran). Shapes: the cfg2 ViT GEMMs at a quarter of the clip (several rounds of the persistent grid,
    ragged last row block), a frame-sharded share (few tiles), the Qwen2 gate/up and down projections at 2 100 rows
    (SwiGLU epilogue; K = 18 944: 296 slabs per tile)."""
    from cogstream_amd import _lib as L2
    ops = _ops()
    g = torch.Generator(device=dev).manual_seed(11)
    rnd = lambda *s: (torch.randn(*s, generator=g, device=dev) * 0.5).bfloat16()

    class _KTile:
        """stands in for the second library of the round-4 form of this test: gemm(..., lib=alt) = the K-tile body"""
    alt = _KTile()
    _gemm = ops.gemm

    def gemm_ab(**kw):
        if kw.pop("lib", None) is None:
            out = _gemm(**kw)
            assert L2.debug_get("gemm_last_body") in (2, 4, 5), L2.debug_get("gemm_last_body")
            return out
        with L2.debug_switch("gemm_pp64", 0):
            out = _gemm(**kw)
            assert L2.debug_get("gemm_last_body") in (2, 3, 5), L2.debug_get("gemm_last_body")
        return out

    class _Ops:
        gemm = staticmethod(gemm_ab)
    ops = _Ops()
    hd, H, I = 72, 1152, 4352
    for M in (59136 // 4 + 40, 6400):
        x, big = rnd(M, H), rnd(M, I)
        nf = hd // 4
        hpos = torch.randint(0, 22, (M,), generator=g, device=dev)
        wpos = torch.randint(0, 42, (M,), generator=g, device=dev)
        inv_freq = 1.0 / (10000.0 ** (torch.arange(nf, dtype=torch.float32, device=dev) / nf))
        ang = torch.cat([hpos[:, None].float() * inv_freq, wpos[:, None].float() * inv_freq], 1)
        table = torch.stack([ang.cos(), ang.sin()], -1).contiguous()
        cases = [
            ("qkv+rope table", dict(a=x, w=rnd(3 * H, H), bias=rnd(3 * H), rope_cos=table, rope_cols=2 * H, head_dim=hd)),
            ("out-proj+res", dict(a=x, w=rnd(H, H), bias=rnd(H), residual=rnd(M, H))),
            ("fc1+gelu", dict(a=x, w=rnd(I, H), bias=rnd(I), act=L2.ACT_GELU_TANH)),
            ("fc2+res", dict(a=big, w=rnd(H, I), bias=rnd(H), residual=rnd(M, H))),
            ("plain", dict(a=x, w=rnd(3584, H))),
            ("K = 128 (two slabs)", dict(a=rnd(M, 128), w=rnd(H, 128))),
        ]
        for name, kw in cases:
            for rep in range(2):
                a = ops.gemm(**kw)
                b = ops.gemm(**kw, lib=alt)
                torch.cuda.synchronize()
                assert torch.equal(a, b), (M, name, rep, float((a.float() - b.float()).abs().max()))
    xq, inter = rnd(2100, 3584), rnd(2100, 18944)
    for name, kw in (("gate/up swiglu", dict(a=xq, w=rnd(2 * 18944, 3584), act=L2.ACT_SWIGLU)),
                     ("down+res", dict(a=inter, w=rnd(3584, 18944), residual=rnd(2100, 3584)))):
        a = ops.gemm(**kw)
        b = ops.gemm(**kw, lib=alt)
        torch.cuda.synchronize()
        assert torch.equal(a, b), (name, float((a.float() - b.float()).abs().max()))


def test_tall_ragged_column_tiles_equal_padded_tiles_bit_for_bit(dev):
    """Round 6: gemm_tn_pp64_kernel walks a ragged column block of <= 128 columns (N = 1152: 4.5 blocks of 256, N = 3456:
    13.5) as tiles of 384 rows x 128 columns through the same LDS ring (csrc/gemm.hip, TALL4) instead of a half-empty 256 x 256
    tile. Every output element keeps its MFMAs and its K order, so the results must be the bits of the padded walk (debug switch
    gemm_tall = 0). Shapes: row counts that are / are not multiples of 384 and of 256, fewer rows than one tall tile's second
    and third row group, a ragged block narrower than 128 columns (generic epilogue in its second column group), every fused
    ViT epilogue of an N = 1152 / 3456 GEMM (bias + residual + row statistics, rotary + head-major), walk groups forced to
    3 / 6 / 9 row blocks, and the few-tile split choices of a frame-sharded share (gemm_split)."""
    from cogstream_amd import _lib as L2
    ops = _ops()
    g = torch.Generator(device=dev).manual_seed(17)
    rnd = lambda *s: (torch.randn(*s, generator=g, device=dev) * 0.5).bfloat16()
    hd, H, I = 72, 1152, 4352

    def both(tag, **kw):
        outs = []
        for tall in (1, 0):
            with L2.debug_switch("gemm_tall", tall):
                rs = kw.get("row_stats")
                if rs is not None:
                    rs.fill_(-7.0)
                o = ops.gemm(**kw).clone()
                body = L2.debug_get("gemm_last_body")
                outs.append((o, None if rs is None else rs.clone(), body))
        torch.cuda.synchronize()
        (a, sa, ba), (b, sb, bb) = outs
        assert torch.equal(a, b), (tag, ba, bb, float((a.float() - b.float()).abs().max()))
        if sa is not None:
            assert torch.equal(sa, sb), (tag, "row statistics")
        return ba

    bodies = set()
    for M in (59136 // 4, 59136 // 4 + 40, 7392, 6400, 1024, 1280 + 8, 1536, 3 * 384 + 130, 2049):
        x, big = rnd(M, H), rnd(M, I)
        nf = hd // 4
        hpos = torch.randint(0, 22, (M,), generator=g, device=dev)
        wpos = torch.randint(0, 42, (M,), generator=g, device=dev)
        inv_freq = 1.0 / (10000.0 ** (torch.arange(nf, dtype=torch.float32, device=dev) / nf))
        ang = torch.cat([hpos[:, None].float() * inv_freq, wpos[:, None].float() * inv_freq], 1)
        table = torch.stack([ang.cos(), ang.sin()], -1).contiguous()
        stats = torch.empty(M, H // 64, 2, device=dev, dtype=torch.float32)
        ln_ab = torch.rand(M, 2, generator=g, device=dev) + 0.5
        col_c = torch.randn(3 * H, generator=g, device=dev)
        cases = [
            ("out-proj + residual + statistics", dict(a=x, w=rnd(H, H), bias=rnd(H), residual=rnd(M, H), row_stats=stats)),
            ("fc2 + residual + statistics", dict(a=big, w=rnd(H, I), bias=rnd(H), residual=rnd(M, H), row_stats=stats)),
            ("qkv + rotary, head-major, LN fold", dict(a=x, w=rnd(3 * H, H), rope_cos=table, rope_cols=2 * H, head_dim=hd,
                                                       ln_ab=ln_ab, col_c=col_c, hm_cols=H)),
            ("qkv + rotary, row-major", dict(a=x, w=rnd(3 * H, H), bias=rnd(3 * H), rope_cos=table, rope_cols=2 * H, head_dim=hd)),
            ("plain, 1.5 column blocks", dict(a=x, w=rnd(384, H))),
            ("plain, ragged block of 76 columns", dict(a=x, w=rnd(1100, H), bias=rnd(1100))),
            ("K = 128 (two slabs)", dict(a=rnd(M, 128), w=rnd(H, 128))),
        ]
        for name, kw in cases:
            bodies.add(both((M, name), **kw))
        for gm in (3, 4, 9):
            with L2.debug_switch("gemm_group_m", gm):
                bodies.add(both((M, "group_m", gm), **cases[0][1]))
        with L2.debug_switch("gemm_split", 0):
            bodies.add(both((M, "unsplit"), **cases[0][1]))
            bodies.add(both((M, "unsplit fc2"), **cases[1][1]))
    assert 4 in bodies, bodies          # the whole-line kernel (and with it the tall walk) did run


@pytest.mark.parametrize("variant", ["deep", "round4", "general", "general hd72"])
def test_prescaled_attention_survives_a_key_far_above_the_running_reference(dev, variant):
    """The deferred-maximum softmax of the pre-scaled kernels evaluates P = exp2(score - reference) against the reference it had
    BEFORE the tile and rescales (O, l) afterwards; a key more than ~127 log2 units above everything a row has seen made P
    infinite and the deferred factor turned it into inf * 0 = NaN (advisor finding, round 5). Such a tile now moves the
    reference in front of its exponentials (PRE_FAR, csrc/attn.hip). One key in the middle of the sequence is scaled until
    its scores reach +-400 log2 units: the output must be finite and as close to the fp32 softmax as without that key."""
    from cogstream_amd import _lib as L2
    ops = _ops()
    hd = 72 if variant.endswith("hd72") else 128
    hq, hkv = (4, 4) if hd == 72 else (28, 4)
    S, far = 1100, 733
    g = torch.Generator().manual_seed(S + hd)
    q = (torch.randn(S, hq * hd, generator=g) * (LOG2E / math.sqrt(hd))).bfloat16()
    k = torch.randn(S, hkv * hd, generator=g).bfloat16()
    v = torch.randn(S, hkv * hd, generator=g).bfloat16()
    k[far] *= 300.0
    big = (q.float().view(S, hq, hd)[far:, :, :] * k.float().view(S, hkv, hd)[far].repeat_interleave(hq // hkv, 0)).sum(-1)
    assert float(big.max()) > 160 and float(big.min()) < -160        # both signs, far beyond fp32's exponent range for exp2
    sw = {"deep": {"attn_prefill_deep": 1}, "round4": {"attn_prefill_deep": 0}, "general": {"attn_prefill_dma": 0},
          "general hd72": {}}[variant]
    causal = hd == 128
    import contextlib
    with contextlib.ExitStack() as st:
        for name, val in sw.items():
            st.enter_context(L2.debug_switch(name, val))
        if hd == 72:
            st.enter_context(L2.debug_switch("attn_vit", 0))
        kw = dict(hq=hq, hkv=hkv, head_dim=hd, causal=causal, q_prescaled=True)
        if hd == 72:
            kw.update(cu_seqlens=torch.tensor([0, S], dtype=torch.int32, device=dev), max_seqlen=S)
        out = ops.attention(q.to(dev), k.to(dev), v.to(dev), **kw)
        want = {"deep": 5, "round4": 5, "general": 1, "general hd72": 1}[variant]
        assert L2.debug_get("attn_last_kernel") == want, L2.debug_get("attn_last_kernel")
    assert torch.isfinite(out.float()).all(), "NaN / inf rows: %d" % int((~torch.isfinite(out.float())).any(1).sum())
    ref = _attn_ref(q, k, v, hq, hkv, hd, causal=causal, scale=math.log(2.0))
    assert rel_err(out.float(), ref) < 1.5e-2, rel_err(out.float(), ref)


def test_round5_prompt_attention_kernel_is_as_close_to_fp32_as_the_round4_kernel(dev):
    """attn_prefill_dma_kernel<1> (round 5, the default) re-orders attn_prefill_dma_kernel<0> (round 4): fragment reads ahead
    of their MFMAs, and the running reference only moves when a score exceeds it by 2^6 (P <= 64 instead of <= 1 before the
    bf16 rounding), so the two outputs differ like two bf16 roundings do; what must hold is that the new kernel is as close to
    the fp32 softmax as the old one. Whole prompts, ragged lengths, two- and three-tile prompts, one-tile sequences in the
    var-len form, prefix-KV continuation (also with fewer than 128 new rows), a late dominant key (the rescale path)."""
    from cogstream_amd import _lib as L2
    ops = _ops()
    g = torch.Generator(device=dev).manual_seed(5)
    hq, hkv, hd = 28, 4, 128
    mk = lambda n, h, sc=0.5: (torch.randn(n, h * hd, generator=g, device=dev) * sc).bfloat16()

    def both(*a, **kw):
        with L2.debug_switch("attn_prefill_deep", 1):
            x = ops.attention(*a, **kw)
            assert L2.debug_get("attn_last_kernel") == 5
        with L2.debug_switch("attn_prefill_deep", 0):
            y = ops.attention(*a, **kw)
            assert L2.debug_get("attn_last_kernel") == 5
        torch.cuda.synchronize()
        assert torch.isfinite(x.float()).all()
        cu = kw.get("cu_seqlens")
        ref = _attn_ref(a[0].cpu(), a[1].cpu(), a[2].cpu(), hq, hkv, hd, cu=None if cu is None else cu.cpu().long(), causal=True,
                        q_pos0=kw.get("q_pos0", 0), scale=math.log(2.0))
        e_new, e_old = rel_err(x.float(), ref), rel_err(y.float(), ref)
        assert e_new < 1.2e-2 and e_new <= 1.5 * e_old + 1e-3, (e_new, e_old)
    for S, pos0 in ((4096, 0), (2100, 0), (2433, 0), (300, 0), (128, 0), (129, 0), (191, 0), (192, 0), (193, 0), (1000, 1500), (130, 700), (127, 173), (5, 40), (64, 0), (63, 0), (2, 0)):
        q, kk, v = mk(S, hq), mk(S + pos0, hkv), mk(S + pos0, hkv, 1.0)
        if S > 2000:
            kk[S - 700] *= 6.0                                                   # the maximum moves late, by a lot
        both(q, kk, v, hq=hq, hkv=hkv, head_dim=hd, causal=True, q_prescaled=True, q_pos0=pos0)
    lens = [300, 1, 64, 129, 1000, 63, 512]
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device=dev)
    S = sum(lens)
    q, kk, v = mk(S, hq), mk(S, hkv), mk(S, hkv, 1.0)
    both(q, kk, v, hq=hq, hkv=hkv, head_dim=hd, causal=True, q_prescaled=True, cu_seqlens=cu, max_seqlen=max(lens))


@pytest.mark.parametrize("M", [300, 1300, 4096])
def test_gemm_row_stats_and_ln_fold(dev, M):
    """LayerNorm fused around the GEMMs: (1) a residual-stream GEMM (N = hidden) also writes per-row partial sums,
    cogs_ln_finalize turns them into (rstd, -rstd * mean); (2) the consuming GEMM reads x itself with W * diag(gamma)
    (rows centred) and applies rstd_r * acc + c_n -- together LN(x) W^T + bias of nn.LayerNorm + nn.Linear
    (model/modeling_videollama3_encoder.py:382-391). M = 300 / 1300 / 4096 reach the 128x128, 256x128 + ragged ping-pong
    and the full ping-pong kernels."""
    ops = _ops()
    torch.manual_seed(M)
    H, N2, eps = 1152, 1472, 1e-6
    a0 = torch.randn(M, 256).bfloat16()
    w0 = (torch.randn(H, 256) / 16).bfloat16()
    b0 = torch.randn(H).bfloat16()
    res = (torch.randn(M, H) * 2 + 0.7).bfloat16()                       # rows with a clearly non-zero mean
    res[::7] += 16.0                                                     # and some with |mean / std| ~ 8
    res[3::11] += 200.0                                                  # and ~ 100 (rows dominated by their mean)
    y32 = a0.float() @ w0.float().t() + b0.float() + res.float()
    part = torch.zeros(M, H // 64, 2, device=dev)
    x = ops.gemm(a0.to(dev), w0.to(dev), b0.to(dev), residual=res.to(dev), row_stats=part)
    assert rel_err(x.float(), y32) < 1.2e-2
    xs = x.float().cpu()                                                 # the statistics are those of the STORED rows
    want = torch.stack([xs.view(M, H // 64, 64).sum(-1), (xs ** 2).view(M, H // 64, 64).sum(-1)], -1)
    assert rel_err(part, want) < 1e-5
    ab = ops.ln_finalize(part, H, eps)
    mean, var = xs.double().mean(1).float(), xs.double().var(1, unbiased=False).float()
    rstd = (var + eps).rsqrt()
    # rows dominated by their mean (|mean / std| ~ 100) lose (mean/std)^2 of the fp32 partial sums' precision in
    # E[x^2] - mean^2: 1e-3 there, 1e-4 everywhere else -- both far below the bf16 rounding of the consumer's output
    huge_rows = torch.zeros(M, dtype=torch.bool)
    huge_rows[3::11] = True
    abc = ab.cpu()
    for rows, tol in ((~huge_rows, 1e-4), (huge_rows, 1e-3)):
        assert rel_err(abc[rows, 0], rstd[rows]) < tol and rel_err(abc[rows, 1], (-rstd * mean)[rows]) < tol
    # consumer: fc1-like (GELU) and plain
    gamma, beta = (1 + 0.3 * torch.randn(H)).bfloat16(), (0.2 * torch.randn(H)).bfloat16()
    w1 = (torch.randn(N2, H) / 30).bfloat16()
    b1 = torch.randn(N2).bfloat16()
    from cogstream_amd.weights import fold_layernorm
    wf, col_s, col_c = fold_layernorm(w1, b1, gamma, beta)               # W * gamma, rows centred AND zero-sum after rounding
    # the stored bf16 rows sum to zero up to one step of their smallest entries (plain rounding leaves ~1e-2 of |w|_1 / K)
    assert float(col_s.abs().max()) < 1e-5 * float(wf.float().abs().sum(1).mean())
    xr = x.float().cpu()
    ln = F.layer_norm(xr, (H,), gamma.float(), beta.float(), eps)
    for act, fn in ((L_ACT_NONE(), lambda t: t), (L_ACT_GELU(), lambda t: F.gelu(t, approximate="tanh"))):
        ref = fn(ln @ w1.float().t() + b1.float())
        out = ops.gemm(x, wf.to(dev), None, act=act, ln_ab=ab, col_c=col_c.to(dev))
        unfused = ops.gemm(ln.bfloat16().to(dev), w1.to(dev), b1.to(dev), act=act)
        # every row class -- ordinary mean, |mean / std| ~ 8 and ~ 100 -- is as accurate as the unfused bf16 path: with
        # zero-sum weight rows the product on x equals the product on x - mean, whatever the mean (round 2 allowed the
        # |mean / std| ~ 8 rows 9x the error: the bf16 rounding of the centred rows left a row sum that the mean multiplied)
        big = torch.zeros(M, dtype=torch.bool)
        big[::7] = True
        huge = torch.zeros(M, dtype=torch.bool)
        huge[3::11] = True
        for rows in (~(big | huge), big & ~huge, huge):
            e_f, e_u = rel_err(out.float().cpu()[rows], ref[rows]), rel_err(unfused.float().cpu()[rows], ref[rows])
            assert e_f < 2.0 * e_u + 2e-3, (act, e_f, e_u)


def test_ln_fold_through_the_c_abi_as_the_header_states_it(dev):
    """cogs_gemm with ln_ab, called through ctypes with operands built by hand from the text of include/cogs.h
    (cogs_gemm_desc.ln_ab): W = rows of W0 * diag(gamma) minus their mean over k, col_c = bias0 + W0 . beta, bias NULL,
    ln_ab[r] = (rstd_r, anything) -- equals nn.Linear(nn.LayerNorm(x)). The second component of ln_ab is filled with
    garbage: the header says it is not read. Unsupported combinations return COGS_E_UNSUPPORTED instead of silently
    dropping the LayerNorm (M == 1 -> GEMV; ln_ab + residual -> no specialised epilogue)."""
    import ctypes as C
    from cogstream_amd import _lib as L
    torch.manual_seed(5)
    M, H, N, eps = 1500, 1152, 640, 1e-6
    x = (torch.randn(M, H) * 1.5 + 0.4).bfloat16()
    gamma, beta = (1 + 0.2 * torch.randn(H)).bfloat16(), (0.1 * torch.randn(H)).bfloat16()
    w0, b0 = (torch.randn(N, H) / 30).bfloat16(), torch.randn(N).bfloat16()
    wg = w0.double() * gamma.double()[None, :]
    w = (wg - wg.mean(dim=1, keepdim=True)).to(torch.bfloat16).contiguous()          # the header's W, rounded once
    col_c = (b0.double() + w0.double() @ beta.double()).float()
    xf = x.float()
    rstd = (xf.var(1, unbiased=False) + eps).rsqrt()
    ln_ab = torch.stack([rstd, torch.full_like(rstd, 1e30)], 1).contiguous()         # [r][1] must not be read
    ref = F.layer_norm(xf, (H,), gamma.float(), beta.float(), eps) @ w0.float().t() + b0.float()
    xd, wd_, cd, abd = x.to(dev), w.to(dev), col_c.to(dev), ln_ab.to(dev)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)

    def desc(m, **kw):
        d = L.GemmDesc()
        d.dtype = L.DT_BF16
        d.A, d.lda, d.W, d.ldw, d.C, d.ldc = xd.data_ptr(), H, wd_.data_ptr(), H, out.data_ptr(), N
        d.M, d.N, d.K = m, N, H
        d.ln_ab, d.col_c = abd.data_ptr(), cd.data_ptr()
        for k, v in kw.items():
            setattr(d, k, v)
        return d

    assert L.lib.cogs_gemm(L.current_stream(), C.byref(desc(M))) == L.OK
    torch.cuda.synchronize()
    assert rel_err(out.float().cpu(), ref) < 1e-2
    assert L.lib.cogs_gemm(L.current_stream(), C.byref(desc(1))) == L.E_UNSUPPORTED
    assert L.lib.cogs_gemm(L.current_stream(), C.byref(desc(M, residual=xd.data_ptr(), ldr=H))) == L.E_UNSUPPORTED
    stats = torch.zeros(M, N // 64, 2, device=dev)
    d = desc(M, act=L.ACT_GELU_ERF)
    d.ln_ab, d.col_c, d.row_stats = None, None, stats.data_ptr()
    assert L.lib.cogs_gemm(L.current_stream(), C.byref(d)) == L.E_UNSUPPORTED        # row_stats + GELU(erf): no such epilogue


@pytest.mark.parametrize("n", [1, 2, 3, 8])
def test_select_near_centroid_vs_torch(dev, n):
    """cogs_select_near_centroid against the reference's per-cluster logic (model/cogreasoner_chat.py:50-64: members if at
    most n, else members[topk(dist, n, largest=False)]) on random distances: clusters with 0, fewer than n, exactly n and
    many members; T beyond one 64-row sweep; exact ties go to the lower row."""
    ops = _ops()
    torch.manual_seed(70 + n)
    T, K = 333, 11
    d2 = torch.rand(T, K)
    d2[5::40] = d2[6::40][: d2[5::40].shape[0]]                                # some exactly equal rows (ties)
    assign = torch.randint(0, K - 3, (T,))
    assign[:2] = K - 3                                                        # a cluster with 2 members
    assign[2:2 + n] = K - 2                                                   # a cluster with exactly n members (cluster K-1: none)
    picks, counts = ops.select_near_centroid(d2.to(dev), assign.to(dev), n)
    picks, counts = picks.cpu(), counts.cpu()
    for k in range(K):
        members = torch.nonzero(assign == k, as_tuple=True)[0]
        if members.numel() <= n:
            want = members
        else:
            dk = d2[members, k]
            order = sorted(range(members.numel()), key=lambda i: (float(dk[i]), int(members[i])))[:n]
            want = members[torch.tensor(order)]
        assert int(counts[k]) == want.numel()
        assert picks[k, :want.numel()].tolist() == want.tolist()
        assert (picks[k, want.numel():] == -1).all()


def L_ACT_NONE():
    from cogstream_amd import _lib as L
    return L.ACT_NONE


def L_ACT_GELU():
    from cogstream_amd import _lib as L
    return L.ACT_GELU_TANH


@pytest.mark.parametrize("M,heads,fold", [(96, 2, False), (1300, 2, True), (1300, 7, True), (2048 + 40, 16, True), (6400, 16, False),
                                          (14824, 16, True)])
def test_gemm_head_major_output_equals_row_major_bit_for_bit(dev, M, heads, fold):
    """cogs_gemm_desc.hm_rows (round 5: the encoder's QKV GEMM stores q | k | v as [block][head][row][hd] so that the
    attention kernel streams whole 128-byte lines): the same epilogue arithmetic, only the store address changes -- the
    head-major result must be the row-major result re-arranged, bit for bit. heads = 2 / 7: hm_cols is not a multiple of
    64, so a wave tile straddles q / k / v and every tile takes the general epilogue; heads = 16 (the production width):
    interior tiles take the lean paired epilogue; M covers the 128x128, 256x128 ring and ping-pong kernels, a ragged last
    row block, and the round-aligned split (14 824 rows)."""
    ops = _ops()
    g = torch.Generator(device=dev).manual_seed(M + heads)
    hd, K = 72, 1152 if heads == 16 else 128
    Hc = heads * hd
    N = 3 * Hc
    rnd = lambda *s: (torch.randn(*s, generator=g, device=dev) * 0.5).bfloat16()
    x = rnd(M, K) + 0.25
    ang = torch.rand(M, hd // 2, generator=g, device=dev) * 6.0
    table = torch.stack([ang.cos(), ang.sin()], -1).contiguous()
    kw = dict(rope_cos=table, rope_sin=None, rope_cols=2 * Hc, head_dim=hd)
    if fold:
        from cogstream_amd.weights import fold_layernorm
        wf, _, col_c = fold_layernorm(rnd(N, K).cpu(), rnd(N).cpu(), (1 + 0.3 * torch.randn(K)).bfloat16(), (0.2 * torch.randn(K)).bfloat16())
        xs = x.float()
        rstd = (xs.var(1, unbiased=False) + 1e-6).rsqrt()
        ab = torch.stack([rstd, -rstd * xs.mean(1)], 1).contiguous()
        args = dict(a=x, w=wf.to(dev), bias=None, ln_ab=ab, col_c=col_c.to(dev), **kw)
    else:
        args = dict(a=x, w=rnd(N, K), bias=rnd(N), **kw)
    row = ops.gemm(**args)
    hm = ops.gemm(**args, hm_cols=Hc)
    torch.cuda.synchronize()
    assert hm.shape == (3, heads, M, hd)
    want = row.view(M, 3, heads, hd).permute(1, 2, 0, 3)
    assert torch.equal(hm, want), float((hm.float() - want.float()).abs().max())


@pytest.mark.parametrize("heads,lens", [(1, [1]), (2, [32, 33]), (1, [63, 64, 65]), (8, [200] * 5), (2, [255, 256, 257]),
                                        (16, [924, 924]), (3, [1024, 70, 1000]), (16, [130] * 4)])
@pytest.mark.parametrize("variant", [2, 1])
def test_attention_head_major_inputs_equal_token_major_bit_for_bit(dev, heads, lens, variant):
    """cogs_attn_desc.head_stride: the encoder's attention kernels (pipelined LDS-DMA kernel = debug switch attn_vit 2,
    unpipelined = 1) reading q, k, v as [head][row][72] -- contiguous key tiles, whole-line LDS-DMA pieces -- must give the
    bits they give on the token-major fused buffer: only the source addresses differ. Ragged last tiles (rows past a
    segment's end are re-read from its last row, never from the next frame or head), one-row segments, both grid orders."""
    from cogstream_amd import _lib as L
    ops = _ops()
    hd, H = 72, heads * 72
    n = sum(lens)
    g = torch.Generator(device=dev).manual_seed(n + heads)
    qkv = torch.randn(n, 3 * H, generator=g, device=dev)
    qkv[:, :H] *= LOG2E / math.sqrt(hd)
    qkv = qkv.bfloat16()
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device=dev)
    with L.debug_switch("attn_vit", variant):
        tok = ops.attention(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], hq=heads, hkv=heads, head_dim=hd, cu_seqlens=cu,
                            max_seqlen=max(lens), q_prescaled=True)
        assert L.debug_get("attn_last_kernel") == (3 if variant == 2 else 2)
        hm3 = qkv.view(n, 3, heads, hd).permute(1, 2, 0, 3).contiguous()           # [3][heads][n][hd], one buffer
        out = ops.attention(hm3[0], hm3[1], hm3[2], hq=heads, hkv=heads, head_dim=hd, cu_seqlens=cu, max_seqlen=max(lens),
                            q_prescaled=True, head_major=True)
        assert L.debug_get("attn_last_kernel") == (8 if variant == 2 else 2)
        # the same with GUARD rows behind every head's rows (head stride = (n + 70) rows): a ragged last tile must repeat the
        # segment's last row, never read on into what follows the head. The guards hold a huge FINITE value (the kernel file is
        # built with -fno-honor-nans, so a NaN is no reliable tracer): one of them taken for a key would own its softmax row,
        # one taken for a value would blow the output up
        G = 70
        guarded = torch.full((3, heads, n + G, hd), 3.0e38, device=dev, dtype=torch.bfloat16)
        guarded[:, :, :n] = hm3
        outg = ops.attention(guarded[0][:, :n], guarded[1][:, :n], guarded[2][:, :n], hq=heads, hkv=heads, head_dim=hd,
                             cu_seqlens=cu, max_seqlen=max(lens), q_prescaled=True, head_major=True)
        assert L.debug_get("attn_last_kernel") == (8 if variant == 2 else 2)
    torch.cuda.synchronize()
    assert torch.isfinite(out.float()).all()
    assert torch.equal(out, tok), float((out.float() - tok.float()).abs().max())
    assert torch.equal(outg, tok), "guard rows behind a head's rows were read: %g" % float((outg.float() - tok.float()).abs().max())
