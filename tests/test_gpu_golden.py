"""HIP path (through the C ABI / host mirror) against the fixtures produced by running the reference
(tests/golden/make_golden.py). Integer outputs -- cluster assignments, selected frames, keep-masks,
compacted ids, generated token ids, selection strings -- must be BIT-EXACT; floating point is compared in
fp32 parity mode at 1e-4 relative (the north-star logit tolerance is 1e-3) and in bf16 at bf16 precision."""
import os
import random

import numpy as np
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
VIT = dict(hidden_size=576, intermediate_size=200, num_hidden_layers=2, num_attention_heads=8)
LLM = dict(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2,
           num_key_value_heads=1, vocab_size=512, image_token_index=258, eos_token_id=257)


def _load(name):
    return {k: v for k, v in np.load(os.path.join(G, name), allow_pickle=False).items()}


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-4), (torch.bfloat16, 3e-2)])
def test_vit_and_projector_vs_reference(dev, dtype, tol):
    from cogstream_amd.vision import BLOCK_DIAG, REF_EAGER_GLOBAL, Projector, VisionEncoder
    from cogstream_amd.weights import VisionConfig, random_proj_state, random_vit_state
    g = _load("vit_tiny.npz")
    cfg = VisionConfig(**VIT)
    enc = VisionEncoder(random_vit_state(cfg, seed=3, std=0.05), cfg, dtype=dtype, device=dev)
    pix = torch.from_numpy(g["pixel_values"]).to(dev)
    grid, merge = torch.from_numpy(g["grid_sizes"]), torch.from_numpy(g["merge_sizes"])
    eg = enc(pix, grid, merge, attn_mode=REF_EAGER_GLOBAL)
    bd = enc(pix, grid, merge, attn_mode=BLOCK_DIAG)
    assert rel_err(eg.float(), torch.from_numpy(g["eager_global"])) < tol
    assert rel_err(bd.float(), torch.from_numpy(g["block_diag"])) < tol
    proj = Projector(random_proj_state(cfg.hidden_size, 256, seed=1, std=0.05), dtype=dtype, device=dev)
    assert rel_err(proj(bd).float(), torch.from_numpy(g["projected"])) < tol


@pytest.mark.parametrize("ci", [0, 1, 2, 3])
def test_kmeans_vs_reference(dev, ci):
    from cogstream_amd.kmeans import kmeans_with_time_min_max, select_additional_frames
    g = _load("kmeans.npz")
    feats = torch.from_numpy(g[f"c{ci}_features"])
    if int(g[f"c{ci}_is_bf16"]):
        feats = feats.bfloat16()
    ts, K, seed = torch.from_numpy(g[f"c{ci}_ts"]), int(g[f"c{ci}_K"]), int(g[f"c{ci}_seed"])
    random.seed(seed)
    torch.manual_seed(seed)
    cf, ct, assign = kmeans_with_time_min_max(feats.to(dev), ts, K)
    assert rel_err(cf.float(), torch.from_numpy(g[f"c{ci}_centres"])) < 1e-5
    assert rel_err(ct, torch.from_numpy(g[f"c{ci}_centre_ts"])) < 1e-6
    if f"c{ci}_assign" in g:
        assert torch.equal(assign.cpu(), torch.from_numpy(g[f"c{ci}_assign"]))          # bit-exact cluster indices
        sel = select_additional_frames(feats.to(dev), cf, assign, 2)
        assert torch.equal(torch.cat(sel).cpu().sort().values, torch.from_numpy(g[f"c{ci}_extra"]))
        assert [len(s) for s in sel] == g[f"c{ci}_extra_counts"].tolist()
    else:
        assert assign is None


def test_kmeans_reseed_branch_vs_reference(dev):
    """The empty-cluster branch (kmeans_with_time.py:116-120) through cogs_kmeans_lloyd: the reseed rows are drawn AHEAD on
    the host into a pool the device consumes in the reference's order, and the Python generator is afterwards put back
    to where the reference leaves it. tests/golden/kmeans_reseed.npz case 0 (the reference on inputs with duplicate
    centres: three clusters reseeded): assignments, centres and the generator position equal the reference's."""
    from cogstream_amd.kmeans import kmeans_with_time_min_max
    g = _load("kmeans_reseed.npz")
    feats, ts, K, seed = torch.from_numpy(g["c0_features"]), torch.from_numpy(g["c0_ts"]), int(g["c0_K"]), int(g["c0_seed"])
    random.seed(seed)
    torch.manual_seed(seed)
    cf, ct, assign = kmeans_with_time_min_max(feats.to(dev), ts, K)
    assert random.random() == float(g["c0_next_random"])              # as many draws consumed as the reference consumed
    assert torch.equal(assign.cpu(), torch.from_numpy(g["c0_assign"]))
    assert rel_err(cf.float(), torch.from_numpy(g["c0_centres"])) < 1e-5
    assert rel_err(ct.cpu() + 1, torch.from_numpy(g["c0_centre_ts"]) + 1) < 1e-6


def test_kmeans_reseed_pool_runs_dry_and_resumes(dev, monkeypatch):
    """Case 1 of the same fixture reseeds seven clusters in EVERY one of the 30 iterations: the pre-drawn pool runs dry
    several times, the library stops without committing the iteration, the host draws more and resumes. Its inputs are
    exact duplicates, so every assignment is an exact tie between duplicate centres -- which the reference resolves
    by the rounding noise of torch.cdist's sgemm form (203 reseeds in the fixture, an order nothing else reproduces:
    DESIGN.md section 2). The yardstick here is therefore the oracle with DIRECT distances (exact ties stay exact, first
    centre wins -- the HIP kernels' arithmetic): same iteration count, number of draws and generator position, through
    at least three refills of the pool, and the same PARTITION of the rows. (Labels may differ: a cluster mean of n
    identical rows is the row only up to the rounding of the n-term sum, whose order differs between torch and the
    kernel, and that last bit decides which of two duplicate centres is nearer.)"""
    from cogstream_amd import kmeans as km
    from cogstream_amd import ops
    from oracle import kmeans as ok
    g = _load("kmeans_reseed.npz")
    feats, ts, K, seed = torch.from_numpy(g["c1_features"]), torch.from_numpy(g["c1_ts"]), int(g["c1_K"]), int(g["c1_seed"])
    random.seed(seed)
    torch.manual_seed(seed)
    want_cf, want_ct, want_assign = ok.kmeans_with_time_min_max(feats, ts, K, exact_distances=True)
    want_next = random.random()
    calls = []
    real = ops.kmeans_lloyd
    monkeypatch.setattr(ops, "kmeans_lloyd", lambda *a: calls.append(real(*a)) or calls[-1])
    # duplicate rows make a k-means++ step with all probabilities zero: the one-call seeding flags it and the product
    # seeds again step by step (after a Lloyd run from the flagged rows that is thrown away) -- count the final run only
    real_step = ops.kmeans_pp_step
    monkeypatch.setattr(ops, "kmeans_pp_step", lambda *a: (calls.clear(), real_step(*a))[1])
    random.seed(seed)
    torch.manual_seed(seed)
    cf, ct, assign = km.kmeans_with_time_min_max(feats.to(dev), ts, K)
    assert random.random() == want_next
    a, b = assign.cpu(), want_assign
    assert torch.equal(a[:, None] == a[None, :], b[:, None] == b[None, :])            # the same partition
    assert torch.equal(torch.bincount(a, minlength=K).sort().values, torch.bincount(b, minlength=K).sort().values)
    for lab in a.unique().tolist():                                                    # each cluster's centre is its rows' value
        rows = feats[a == lab].reshape(-1, cf[0].numel())
        assert rel_err(cf[lab].reshape(-1).cpu(), rows[0]) < 1e-5
    assert km.last_stats["kpp_path"].startswith("step by step")
    assert km.last_stats["iterations"] == 30 == sum(c[0] for c in calls)
    assert sum(1 for c in calls if c[2]) >= 3 and sum(c[1] for c in calls) == 210       # 30 iterations x 7 empty clusters


def _tiny_model(dev, dtype, attn_mode):
    from cogstream_amd.chat import CogReasoner
    from cogstream_amd.llm import Qwen2Engine
    from cogstream_amd.vision import Projector, VisionEncoder
    from cogstream_amd.weights import LlmConfig, VisionConfig, random_llm_state, random_proj_state, random_vit_state
    vcfg, lcfg = VisionConfig(**VIT), LlmConfig(**LLM)
    enc = VisionEncoder(random_vit_state(vcfg, seed=3, std=0.05), vcfg, dtype=dtype, device=dev, attn_mode=attn_mode)
    proj = Projector(random_proj_state(vcfg.hidden_size, lcfg.hidden_size, seed=1, std=0.05), dtype=dtype, device=dev)
    eng = Qwen2Engine(random_llm_state(lcfg, seed=7, std=0.05), lcfg, dtype=dtype, device=dev)
    # like the reference's freshly built tiny model: default GenerationConfig (greedy, no penalty); the answer
    # generation passes repetition_penalty=1.05 explicitly, exactly as make_golden.py does
    return CogReasoner(enc, proj, eng, lcfg, generation_config=dict(do_sample=False, eos_token_id=[257]))


def test_compression_vs_reference(dev):
    g = _load("compress.npz")
    model = _tiny_model(dev, torch.float32, 0)
    pix = torch.from_numpy(g["pixel_values"])
    grid, merge = torch.from_numpy(g["grid_sizes"]), torch.from_numpy(g["merge_sizes"])
    batched = grid.prod(dim=1).div(merge ** 2).long()
    for tag, px in (("f32", pix), ("bf16", pix.bfloat16())):
        m = model._get_compression_mask(px.to(dev), batched, grid, merge, ["video"], minor_frame_indices=[])
        assert torch.equal(m.cpu(), torch.from_numpy(g[f"mask_{tag}"]))
        m = model._get_compression_mask(px.to(dev), batched, grid, merge, ["video"], minor_frame_indices=[2, 5])
        assert torch.equal(m.cpu(), torch.from_numpy(g[f"mask_minor_{tag}"]))
    ev = model.compress_unimportant_events(torch.from_numpy(g["mm"]).to(dev), 6, [1, 4])
    assert rel_err(ev, torch.from_numpy(g["event_pooled"])) < 1e-6
    rows, ids2, _ = model._compress_visual_tokens(torch.from_numpy(g["mask_minor_f32"]), torch.from_numpy(g["input_ids"]), None)
    assert torch.equal(ids2, torch.from_numpy(g["ids_compressed"]))
    assert torch.equal(torch.from_numpy(g["event_pooled"])[rows], torch.from_numpy(g["mm_compressed"]))


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_end_to_end_vs_reference(dev, tag):
    """qa_selection -> generate, fp32 parity mode with the reference's CPU attention semantics
    (REF_EAGER_GLOBAL), toy tokenizer, greedy. a: 8 frames + history (retrieval, prompt surgery);
    b: 150 frames (k-means, 10 event passes); c: b with forced event/question cosines (event compression)."""
    from golden.inputs import FORCED_COSINE, e2e_inputs
    from toy_tokenizer import ToyTokenizer
    g = _load("e2e.npz")
    inp = e2e_inputs(tag)
    assert abs(float(inp["pixel_values"].double().abs().sum()) - float(g[f"{tag}_pix_checksum"])) < 1e-6
    model = _tiny_model(dev, torch.float32, 1)
    tok = ToyTokenizer()
    enc = tok(inp["text"])
    random.seed(5)
    torch.manual_seed(5)
    sel = model.qa_selection(current_question=inp["current_question"], hist_qs=inp["hist_qs"], hist_as=inp["hist_as"],
                             tokenizer=tok, original_text=inp["text"], input_ids=enc["input_ids"],
                             attention_mask=enc["attention_mask"], mode="FCC", all_timestamps=inp["timestamps"])
    assert sel["selection_module_output"] == str(g[f"{tag}_selection"])
    assert sel["if_visual"] == bool(g[f"{tag}_if_visual"])
    assert torch.equal(sel["new_input_ids"][0], torch.from_numpy(g[f"{tag}_new_input_ids"]))
    model.cosine_override = FORCED_COSINE if tag == "c" else None
    ids, sel_out = model.generate(pixel_values=inp["pixel_values"], grid_sizes=inp["grid_sizes"],
                                  merge_sizes=inp["merge_sizes"], modals=["video"], new_input_ids=sel["new_input_ids"],
                                  new_attention_mask=sel["new_attention_mask"],
                                  selection_module_output=sel["selection_module_output"], if_visual=sel["if_visual"],
                                  total_image_num=inp["T"], max_new_tokens=8, repetition_penalty=1.05)
    dbg = model.last_debug
    assert dbg["minor_frames"] == g[f"{tag}_minor"].tolist()                           # bit-exact frame indices
    assert torch.equal(dbg["compression_mask"].cpu(), torch.from_numpy(g[f"{tag}_mask"]))
    if tag in ("b", "c"):
        assert dbg["assign"] == g[f"{tag}_assign"].tolist()                            # bit-exact cluster indices
        assert rel_err(dbg["cosine_raw"], torch.from_numpy(g[f"{tag}_cosine"])) < 1e-3
    assert ids.shape[0] == 1 and ids[0].tolist() == g[f"{tag}_tokens"].tolist()        # greedy tokens
    assert sel_out == str(g[f"{tag}_sel_out"])


def test_end_to_end_bf16_block_diag_runs(dev):
    """production configuration (bf16, per-frame attention): same control flow, integer products stay sane"""
    from golden.inputs import e2e_inputs
    from toy_tokenizer import ToyTokenizer
    inp = e2e_inputs("b")
    model = _tiny_model(dev, torch.bfloat16, 0)
    tok = ToyTokenizer()
    enc = tok(inp["text"])
    random.seed(5)
    torch.manual_seed(5)
    sel = model.qa_selection(current_question=inp["current_question"], hist_qs=[], hist_as=[], tokenizer=tok,
                             original_text=inp["text"], input_ids=enc["input_ids"], attention_mask=enc["attention_mask"],
                             mode="FCC", all_timestamps=inp["timestamps"])
    ids, _ = model.generate(pixel_values=inp["pixel_values"].bfloat16(), grid_sizes=inp["grid_sizes"],
                            merge_sizes=inp["merge_sizes"], modals=["video"], new_input_ids=sel["new_input_ids"],
                            new_attention_mask=sel["new_attention_mask"], if_visual=True, total_image_num=inp["T"],
                            max_new_tokens=4)
    assert ids.shape == (1, 4) or ids.shape[1] <= 4
    assert len(model.last_debug["assign"]) == 150 and int(model.last_debug["compression_mask"].sum()) > 150


def test_multi_turn_session_runs(dev):
    """cfg4-style streaming session on the tiny model: 3 segments, growing history, retrieval from turn 2 on"""
    from cogstream_amd import processing as pr
    from cogstream_amd.answer_generate import run_session, shard_videos
    from toy_tokenizer import ToyTokenizer
    model = _tiny_model(dev, torch.bfloat16, 0)
    tok = ToyTokenizer()
    proc = pr.CogStreamProcessor(tok)
    segs = []
    for i in range(3):
        fr, ts = pr.synthetic_clip(4, 56, 56, kind="drift", clip_idx=i)
        segs.append({"video": fr, "timestamps": [t + 4 * i for t in ts], "questions": [f"What happens in part {i}?"] + (["And then?"] if i == 1 else [])})
    recs = run_session(model, proc, segs, max_new_tokens=4, do_sample=False)
    assert [r["qa_id"] for r in recs] == [0, 1, 2, 3]
    assert recs[0]["predicted_coi"] == [] and len(recs[3]["predicted_coi"]) == 3
    assert all(isinstance(r["prediction"], str) for r in recs)
    assert shard_videos(10, 1, 4) == [1, 5, 9] and shard_videos(10, 3, 4) == [3, 7, 1]


def test_visual_token_cache_is_transparent(dev):
    """section 8f rank 3: the same streaming session with the visual-token cache on (and GPU pre-processing) must
    produce the same records as without; later turns hit the cache for the segments already seen"""
    from cogstream_amd import processing as pr
    from cogstream_amd.answer_generate import run_session
    from toy_tokenizer import ToyTokenizer
    tok = ToyTokenizer()
    segs = []
    for i in range(3):
        fr, ts = pr.synthetic_clip(4, 56, 56, kind="drift", clip_idx=i)
        segs.append({"video": fr, "timestamps": [t + 4 * i for t in ts], "questions": [f"What happens in part {i}?"]})
    recs = []
    for cached in (False, True):
        model = _tiny_model(dev, torch.bfloat16, 0)
        if cached:
            model.enable_visual_cache()
        random.seed(3)
        torch.manual_seed(3)
        recs.append(run_session(model, pr.CogStreamProcessor(tok, device=dev), segs, max_new_tokens=4, do_sample=False))
    assert [r["prediction"] for r in recs[0]] == [r["prediction"] for r in recs[1]]
    assert [r["predicted_coi"] for r in recs[0]] == [r["predicted_coi"] for r in recs[1]]
    assert model.visual_cache_stats == {"hits": 3, "misses": 3}      # turn t sees t+1 segments: 1+2+3 = 6 lookups


def test_full_size_kmeans_and_mask_match_oracle(dev):
    """BASELINE.json configs[2] scale (256 frames, features [256, 50 x 3584] bf16, K = 18; pixel values of 64
    frames at 308x588): cluster assignments, near-centroid frame picks and the pixel-difference keep-mask of the
    HIP path are BIT-equal to the oracle's on the same inputs"""
    from oracle import compress as oc
    from oracle import kmeans as ok
    from cogstream_amd import ops
    from cogstream_amd.kmeans import kmeans_with_time_min_max, select_additional_frames
    T, P, D, K = 256, 50, 3584, 18
    g = torch.Generator().manual_seed(11)
    centres = torch.randn(K, 1, D, generator=g)
    which = torch.arange(T) * K // T
    feats = (centres[which] + 0.35 * torch.randn(T, P, D, generator=g)).to(torch.bfloat16)     # clustered synthetic data
    ts = torch.arange(T, dtype=torch.float32)
    random.seed(0)
    torch.manual_seed(0)
    cf, ct, assign = kmeans_with_time_min_max(feats.to(dev), ts, K)
    picks = select_additional_frames(feats.to(dev), cf, assign, 2)
    random.seed(0)
    torch.manual_seed(0)
    ocf, oct_, oassign = ok.kmeans_with_time_min_max(feats, ts, K)
    opicks = ok.select_additional_frames(feats, ocf, oassign, 2)
    assert torch.equal(assign.cpu(), oassign)
    assert sorted(torch.cat(picks).cpu().tolist()) == sorted(torch.cat(opicks).tolist())
    # pixel-difference mask at cfg2 size
    Tp, gh, gw = 64, 22, 42
    pix = (torch.randn(Tp * gh * gw, 588, generator=g) * 0.4)
    pix[gh * gw:] = pix[:-gh * gw] * 0.5 + pix[gh * gw:] * 0.5                               # correlate neighbouring frames
    pix = pix.to(torch.bfloat16)
    grid, merge = torch.tensor([[Tp, gh, gw]]), torch.tensor([2])
    minor = list(range(0, Tp, 7))
    md = torch.zeros(Tp, dtype=torch.uint8, device=dev)
    md[minor] = 1
    mask = ops.pixdiff_mask(pix.to(dev), Tp, gh * gw // 4, 0.1, 1, md)
    om = oc.compression_mask(pix, grid, merge, ["video"], minor_frame_indices=minor)
    assert torch.equal(om, mask.cpu().bool()) and 0 < int(mask.sum()) < mask.numel()


def test_edge_cases_single_frame_tiny_frame_and_text_only(dev):
    """Edge cases of the product path on the tiny model: a one-frame video (the reference keeps every token of a
    t == 1 video, model/cogreasoner_chat.py:400-403), a 20 x 30 frame that the token floor upsizes, static frames
    (pixel-diff keeps one token per later frame, :413-414), and a text-only question (generate() embeds the ids
    directly, :798-800) -- which must equal the engine driven by hand."""
    import numpy as np
    from cogstream_amd import processing as pr
    from cogstream_amd.answer_generate import infer
    from toy_tokenizer import ToyTokenizer
    model = _tiny_model(dev, torch.bfloat16, 0)
    tok = ToyTokenizer()
    proc = pr.CogStreamProcessor(tok, device=dev)
    one, ts1 = pr.synthetic_clip(1, 20, 30, kind="noise", clip_idx=1)
    tiny = np.random.default_rng(3).integers(0, 255, (2, 20, 30, 3), dtype=np.uint8)
    static = np.repeat(pr.synthetic_clip(1, 20, 30, kind="noise", clip_idx=2)[0], 5, axis=0)
    conv = [{"role": "user", "content": [{"type": "video", "video": one, "timestamps": ts1},
                                         {"type": "video", "video": tiny, "timestamps": [1.0, 2.0]},
                                         {"type": "video", "video": static, "timestamps": [3.0, 4.0, 5.0, 6.0, 7.0]},
                                         {"type": "text", "text": "What is shown?"}]}]
    inputs = proc(conversation=conv, add_system_prompt=True, add_generation_prompt=True, return_tensors="pt")
    grids = inputs["grid_sizes"].tolist()
    assert [g[0] for g in grids] == [1, 2, 5] and all(g[1] % 2 == 0 and g[2] % 2 == 0 for g in grids)
    assert all(g[1] * g[2] // 4 >= 16 for g in grids)                         # min_tokens floor per frame
    out, sel = infer(conv, model, proc, max_new_tokens=3, do_sample=False)
    mask = model.last_debug["compression_mask"].cpu()
    per = [g[1] * g[2] // 4 for g in grids]
    a, b = per[0], per[0] + 2 * per[1]
    assert bool(mask[:a].all())                                                # one-frame video: everything kept
    st = mask[b:].view(5, per[2])
    assert bool(st[0].all()) and st[1:].sum(dim=1).tolist() == [1, 1, 1, 1]    # static frames: one token each
    assert isinstance(out, str) and sel == ""
    # frames of different sizes in one conversation: the reference asserts that every frame has the same number of
    # tokens (:549), and so does this path
    big, tsb = pr.synthetic_clip(2, 112, 224, kind="noise", clip_idx=4)      # 32 tokens per frame against 16
    conv_bad = [{"role": "user", "content": [{"type": "video", "video": static[:3], "timestamps": [0.0, 1.0, 2.0]},
                                             {"type": "video", "video": big, "timestamps": [3.0, 4.0]},
                                             {"type": "text", "text": "What is shown?"}]}]
    with pytest.raises(AssertionError):
        infer(conv_bad, model, proc, max_new_tokens=2, do_sample=False)
    # text-only turn
    conv_t = [{"role": "user", "content": "Hello there, what can you do?"}]
    it = proc(conversation=conv_t, add_system_prompt=True, add_generation_prompt=True, return_tensors="pt")
    assert "pixel_values" not in it and it["total_image_num"] == 0
    sel_in = model.qa_selection(**it, mode="FCC")
    ids, _ = model.generate(**sel_in, max_new_tokens=5, do_sample=False)
    want = model.llm.generate(model.llm.embed_tokens(it["input_ids"].reshape(-1)), max_new_tokens=5, eos_token_id=[257])
    assert ids[0].tolist() == want


def _dist(a, b):
    """(max-norm, RMS) distance of a from b, both relative to b, in fp64"""
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    d = a - b
    return float(d.abs().max() / b.abs().max()), float(d.pow(2).mean().sqrt() / b.pow(2).mean().sqrt())


def _as_good_as_reference_bf16(hip16, ref16, ref32, what, slack=1.5):
    """The production bf16 path is held to the reference's OWN bf16 error: its distance from the reference's fp32
    result may not exceed `slack` x the distance of the reference run in bf16 (CPU, evaluate/answer_generate.py:176
    cast) from the same fp32 result, in max-norm AND in RMS; and it must sit within that same radius of the bf16
    reference itself (two bf16 evaluations with different fusion / summation order cannot be closer than that)."""
    rm, rr = _dist(ref16, ref32)
    hm, hr = _dist(hip16, ref32)
    bm, br = _dist(hip16, ref16)
    print(f"{what}: reference bf16 vs fp32 max {rm:.2e} rms {rr:.2e} | HIP bf16 vs fp32 max {hm:.2e} rms {hr:.2e} | "
          f"HIP bf16 vs reference bf16 max {bm:.2e} rms {br:.2e}")
    assert hm <= slack * rm and hr <= slack * rr, (what, hm, rm, hr, rr)
    assert bm <= 2 * slack * rm and br <= 2 * slack * rr, (what, bm, rm, br, rr)


@pytest.mark.parametrize("tag", ["t", "g"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_qwen2_vs_reference_fixture(dev, tag, dtype):
    """cogs_llm_forward against vectors the REFERENCE's model object produced (tests/golden/qwen2.npz):
    last_hidden_state, sequence mean, prefill logits, three cached decode-step logits. fp32 parity mode: 1e-3 of
    max (north_star's logit tolerance; measured ~1e-5). bf16 production mode: against the reference run in bf16."""
    import json
    from cogstream_amd.llm import Qwen2Engine
    from cogstream_amd.weights import LlmConfig, random_llm_state
    g = _load("qwen2.npz")
    lcfg = LlmConfig(**json.loads(str(g[f"{tag}_cfg"])))
    lst = random_llm_state(lcfg, seed=7, std=0.05)
    assert abs(float(sum(v.double().abs().sum() for v in lst.values())) - float(g[f"{tag}_llm_checksum"])) < 1e-6
    eng = Qwen2Engine(lst, lcfg, dtype=dtype, device=dev)
    emb = torch.from_numpy(g[f"{tag}_embeds"]).to(dev, dtype)
    S = emb.shape[0]
    cache = eng.new_cache(S + 4)
    res = eng.forward(emb, cache, want_pooled=True, want_hidden=True)
    lg = [res["logits"].clone()]
    for tok in g[f"{tag}_step_tokens"].tolist():
        r = eng.forward(eng.embed_tokens(torch.tensor([tok])), cache)
        lg.append(r["logits"].clone())
    lg = torch.stack(lg)
    ref32 = {k: torch.from_numpy(g[f"{tag}_{k}_f32"]) for k in ("hidden", "pooled", "logits")}
    got = {"hidden": res["hidden"].float(), "pooled": res["pooled"], "logits": lg}
    if dtype == torch.float32:
        for k in got:
            m, r = _dist(got[k], ref32[k])
            print(f"qwen2[{tag}] fp32 {k}: max {m:.2e} rms {r:.2e}")
            assert m < 1e-4 and r < 1e-4, (k, m, r)          # north_star asks 1e-3; measured 6e-6
        assert lg.argmax(dim=1)[:3].tolist() == g[f"{tag}_step_tokens"].tolist()
    else:
        for k in got:
            _as_good_as_reference_bf16(got[k], torch.from_numpy(g[f"{tag}_{k}_bf16"]), ref32[k], f"qwen2[{tag}] {k}")


def test_vit_bf16_vs_reference_bf16_fixture(dev):
    """production ViT + projector (bf16, block-diagonal, pre-scaled Q, deferred-max softmax) against the reference
    encoder cast to bf16 and called one frame at a time (tests/golden/vit_tiny_bf16.npz)."""
    from cogstream_amd.vision import BLOCK_DIAG, Projector, VisionEncoder
    from cogstream_amd.weights import VisionConfig, random_proj_state, random_vit_state
    g, g16 = _load("vit_tiny.npz"), _load("vit_tiny_bf16.npz")
    cfg = VisionConfig(**VIT)
    enc = VisionEncoder(random_vit_state(cfg, seed=3, std=0.05), cfg, dtype=torch.bfloat16, device=dev)
    pix = torch.from_numpy(g["pixel_values"]).bfloat16().to(dev)
    grid, merge = torch.from_numpy(g["grid_sizes"]), torch.from_numpy(g["merge_sizes"])
    bd = enc(pix, grid, merge, attn_mode=BLOCK_DIAG)
    _as_good_as_reference_bf16(bd.float(), torch.from_numpy(g16["block_diag"]), torch.from_numpy(g["block_diag"]), "vit tokens")
    proj = Projector(random_proj_state(cfg.hidden_size, 256, seed=1, std=0.05), dtype=torch.bfloat16, device=dev)
    _as_good_as_reference_bf16(proj(bd).float(), torch.from_numpy(g16["projected"]), torch.from_numpy(g["projected"]), "projected")


def test_sampling_warpers_vs_transformers_fixture(dev):
    """cogs_logits_process + cogs_sample against transformers' RepetitionPenalty / Temperature / TopK / TopP
    processors and torch.multinomial on the CPU generator (tests/golden/sampling.npz): the surviving id SET is
    exact, renormalised probabilities agree to fp32 rounding, and with the generator's exponential draws handed
    over the sampled id is the reference sampler's for every seed. Cases: the shipped config on peaked / flat /
    tied rows, no top-k (full-row top-p), top-k only, a one-survivor row, top_p 0.95 over the whole row."""
    from cogstream_amd import ops
    g = _load("sampling.npz")
    for ci in range(int(g["n_cases"])):
        top_k, top_p, temp, rep = g[f"s{ci}_params"].tolist()
        lg = torch.from_numpy(g[f"s{ci}_logits"]).to(dev).clone()
        prev = torch.from_numpy(g[f"s{ci}_prev"]).to(dev)
        ops.logits_process(lg, prev if rep != 1.0 else None, rep, None, 1.0)
        assert torch.equal(lg.cpu(), torch.from_numpy(g[f"s{ci}_after_penalty"]))
        tok, kid, kp = ops.sample(lg, int(top_k), top_p, seed=1, offset=0, want_kept=True, temperature=temp)
        order = kid.cpu().argsort()
        assert kid.cpu()[order].tolist() == g[f"s{ci}_kept"].tolist(), ci                 # exact survivor set
        assert torch.allclose(kp.cpu()[order], torch.from_numpy(g[f"s{ci}_probs"]), rtol=2e-6, atol=1e-9), ci
        assert int(tok) in g[f"s{ci}_kept"].tolist()
        for seed, want in enumerate(g[f"s{ci}_draws"].tolist()):
            q = torch.empty(lg.numel()).exponential_(1, generator=torch.Generator().manual_seed(1000 + seed))
            assert int(ops.sample(lg, int(top_k), top_p, draws=q.to(dev), temperature=temp)) == want, (ci, seed)
    # device draws: the empirical distribution over many Philox offsets follows the kept probabilities
    lg = torch.from_numpy(g["s1_logits"]).to(dev).clone()
    ops.logits_process(lg, torch.from_numpy(g["s1_prev"]).to(dev), 1.05, None, 1.0)
    out = torch.empty(4000, dtype=torch.int64, device=dev)
    for i in range(4000):
        ops.sample(lg, 20, 0.8, seed=99, offset=i, out=out[i:i + 1], temperature=0.7)
    ids, probs = g["s1_kept"].tolist(), g["s1_probs"]
    cnt = torch.bincount(out.cpu(), minlength=lg.numel())
    assert int(cnt.sum()) == int(cnt[ids].sum())                                              # only survivors
    emp = cnt[ids].double() / 4000
    assert float((emp - torch.from_numpy(probs).double()).abs().max()) < 4 * (0.25 / 4000) ** 0.5


def test_sampled_generation_equals_reference_cpu_sampler(dev):
    """the reference pipeline with do_sample=True (its shipped generation mode, model/generation_config.json:2-12)
    on the tiny model, CPU generator seeded: sampler='host' makes every step's multinomial draw from the same
    generator stream, so the 12 sampled ids equal the reference's (fp32 parity mode, eager-global attention)."""
    from golden.inputs import e2e_inputs
    from toy_tokenizer import ToyTokenizer
    g = _load("sampling.npz")
    gc = __import__("json").loads(str(g["generation_config"]))
    inp = e2e_inputs("a")
    model = _tiny_model(dev, torch.float32, 1)
    tok = ToyTokenizer()
    enc = tok(inp["text"])
    sel = model.qa_selection(current_question=inp["current_question"], hist_qs=[], hist_as=[], tokenizer=tok,
                             original_text=inp["text"], input_ids=enc["input_ids"], attention_mask=enc["attention_mask"],
                             mode="NC", all_timestamps=inp["timestamps"])
    for seed in (11, 12):
        random.seed(seed)
        torch.manual_seed(seed)
        ids, _ = model.generate(pixel_values=inp["pixel_values"], grid_sizes=inp["grid_sizes"], merge_sizes=inp["merge_sizes"],
                                modals=["video"], new_input_ids=sel["new_input_ids"], new_attention_mask=sel["new_attention_mask"],
                                selection_module_output=sel["selection_module_output"], if_visual=sel["if_visual"],
                                total_image_num=inp["T"], max_new_tokens=12, do_sample=True,
                                temperature=float(g["gen_temperature"]), top_k=gc["top_k"], top_p=gc["top_p"],
                                repetition_penalty=gc["repetition_penalty"], sampler="host")
        assert ids[0].tolist() == g[f"gen_seed{seed}"].tolist(), seed
    # production sampler: tokens stay on the device, reproducible from the seed
    a, _ = model.generate(pixel_values=inp["pixel_values"], grid_sizes=inp["grid_sizes"], merge_sizes=inp["merge_sizes"],
                          modals=["video"], new_input_ids=sel["new_input_ids"], new_attention_mask=sel["new_attention_mask"],
                          if_visual=True, total_image_num=inp["T"], max_new_tokens=12, do_sample=True, temperature=2.8,
                          top_k=20, top_p=0.8, repetition_penalty=1.05, seed=5)
    b, _ = model.generate(pixel_values=inp["pixel_values"], grid_sizes=inp["grid_sizes"], merge_sizes=inp["merge_sizes"],
                          modals=["video"], new_input_ids=sel["new_input_ids"], new_attention_mask=sel["new_attention_mask"],
                          if_visual=True, total_image_num=inp["T"], max_new_tokens=12, do_sample=True, temperature=2.8,
                          top_k=20, top_p=0.8, repetition_penalty=1.05, seed=5)
    assert a.tolist() == b.tolist() and a.shape[1] <= 12


def test_from_pretrained_equals_the_directly_built_model(dev, tmp_path):
    """CogReasoner.from_pretrained (evaluate/answer_generate.py:173-183) on a synthesised checkpoint directory of the
    tiny e2e model: config + index + 3 shards streamed to HBM, every tensor consumed once; the loaded model
    reproduces the REFERENCE's greedy tokens of e2e case 'a' (fp32 + eager = the reference's CPU semantics), picks up
    generation_config.json, and a peft adapter directory loads into a merged weight set."""
    import json
    from safetensors.torch import save_file
    from golden.inputs import e2e_inputs
    from toy_tokenizer import ToyTokenizer
    from cogstream_amd import checkpoint as ck
    from cogstream_amd.chat import CogReasoner
    from cogstream_amd.weights import (LlmConfig, VisionConfig, random_llm_state, random_lora_state, random_proj_state,
                                       random_vit_state)
    g = _load("e2e.npz")
    vcfg, lcfg = VisionConfig(**VIT), LlmConfig(**LLM)
    d = str(tmp_path / "ckpt")
    ck.save_checkpoint(d, random_vit_state(vcfg, seed=3, std=0.05), random_proj_state(vcfg.hidden_size, lcfg.hidden_size, seed=1, std=0.05),
                       random_llm_state(lcfg, seed=7, std=0.05), vcfg, lcfg, generation={"do_sample": False, "eos_token_id": [257]},
                       n_shards=3, dtype=torch.float32)
    model = CogReasoner.from_pretrained(d, torch_dtype=torch.float32, attn_implementation="eager", device=dev)
    assert model.generation_config["do_sample"] is False and model.generation_config["eos_token_id"] == [257]
    assert model.to(dev.index) is model and model.eval() is model
    with pytest.raises(ValueError):
        CogReasoner.from_pretrained(d, attn_implementation="sdpa", device=dev)       # broken in the reference too
    inp = e2e_inputs("a")
    tok = ToyTokenizer()
    enc = tok(inp["text"])
    sel = model.qa_selection(current_question=inp["current_question"], hist_qs=inp["hist_qs"], hist_as=inp["hist_as"],
                             tokenizer=tok, original_text=inp["text"], input_ids=enc["input_ids"],
                             attention_mask=enc["attention_mask"], mode="FCC", all_timestamps=inp["timestamps"])
    assert sel["selection_module_output"] == str(g["a_selection"])
    ids, _ = model.generate(pixel_values=inp["pixel_values"], grid_sizes=inp["grid_sizes"], merge_sizes=inp["merge_sizes"],
                            modals=["video"], new_input_ids=sel["new_input_ids"], new_attention_mask=sel["new_attention_mask"],
                            selection_module_output=sel["selection_module_output"], if_visual=sel["if_visual"],
                            total_image_num=inp["T"], max_new_tokens=8, repetition_penalty=1.05)
    assert ids[0].tolist() == g["a_tokens"].tolist()
    # bf16 load of the same directory (the reference's torch_dtype=torch.bfloat16) runs the production kernels
    m16 = CogReasoner.from_pretrained(d, torch_dtype=torch.bfloat16, device=dev)
    assert m16.dtype == torch.bfloat16 and m16.vision_encoder.attn_mode == 0
    # adapter directory -> merged weight set, switchable like peft's set_adapter
    ad = str(tmp_path / "adapter")
    os.makedirs(ad)
    lora = random_lora_state(lcfg, r=4, proj_dims=(vcfg.hidden_size, lcfg.hidden_size))
    save_file({k: v.contiguous() for k, v in lora.items()}, os.path.join(ad, "adapter_model.safetensors"))
    json.dump({"r": 4, "lora_alpha": 8}, open(os.path.join(ad, "adapter_config.json"), "w"))
    model.load_adapter_from_path(ad, "full_module")
    model.set_adapter("full_module")
    emb = model.llm.embed_tokens(enc["input_ids"].reshape(-1))
    a = model.llm.forward(emb)["logits"].clone()
    model.set_adapter("base")
    b = model.llm.forward(emb)["logits"]
    assert rel_err(a, b) > 1e-3                                                       # the adapter changes the logits


def test_kmeans_beyond_the_old_caps_matches_oracle(dev):
    """a 600-frame session (three 180-frame segments and more): K = ceil(600/15) = 40 > 32 clusters, and T = 1500 rows
    > 1024 -- the reference has no limit (model/kmeans_with_time.py); assignments and near-centroid picks bit-equal
    to the oracle's"""
    from oracle import kmeans as ok
    from cogstream_amd.kmeans import kmeans_with_time_min_max, select_additional_frames
    for T, P, D, K, bf in ((600, 4, 96, 40, True), (1500, 2, 64, 100, False)):
        g = torch.Generator().manual_seed(T)
        centres = torch.randn(K, 1, D, generator=g) * 3
        which = torch.arange(T) * K // T
        feats = centres[which] + 0.4 * torch.randn(T, P, D, generator=g)
        if bf:
            feats = feats.bfloat16()
        ts = torch.arange(T, dtype=torch.float32)
        random.seed(1); torch.manual_seed(1)
        cf, ct, assign = kmeans_with_time_min_max(feats.to(dev), ts, K)
        picks = select_additional_frames(feats.to(dev), cf, assign, 2)
        random.seed(1); torch.manual_seed(1)
        ocf, oct_, oassign = ok.kmeans_with_time_min_max(feats, ts, K)
        opicks = ok.select_additional_frames(feats, ocf, oassign, 2)
        assert torch.equal(assign.cpu(), oassign), (T, K)
        assert rel_err(cf.float(), ocf) < 1e-5 and rel_err(ct, oct_) < 1e-6
        assert [sorted(p.cpu().tolist()) for p in picks] == [sorted(p.tolist()) for p in opicks]


def test_kmeans_near_ties_follow_the_exact_distance(dev):
    """tests/golden/kmeans_ties.npz (tests/golden/kmeans_tie_study.py ran the reference on these 400 rows of real
    width 179 200, planted at relative margins 1e-2 .. 1e-6 between two centres): the HIP assignment is the EXACT
    (fp64) nearest centre for every row; it equals the reference's for every row whose margin is >= 1e-3, which is
    where the reference's own |x|^2+|c|^2-2x.c sgemm still decides correctly (below that it mis-assigns 3-40 % of
    the rows against the exact distance: the table in DESIGN.md section 2)."""
    from golden.tie_inputs import K, P, D, tie_inputs
    from cogstream_amd import ops
    g = _load("kmeans_ties.npz")
    x = tie_inputs()["features"].view(-1, P * D)
    assert abs(float(x.double().abs().sum()) - float(g["checksum"])) < 1e-3
    T = x.shape[0]
    xd = x.to(dev)
    ws = ops.kmeans_workspace(T, P * D, K, dev)
    d2 = ops.kmeans_sqdist(xd, xd[:K].contiguous(), None, K, ws)
    zeros = torch.zeros(T, device=dev)
    assign, counts = ops.kmeans_assign(d2, zeros, torch.zeros(K, device=dev), 2.0)
    assign = assign.cpu().numpy()
    assert int(counts.sum()) == T
    assert (assign == g["exact_assign"]).all()
    safe = g["margin"] >= 1e-3
    assert safe.sum() > 60 and (assign[safe] == g["ref_assign_8t"][safe]).all()
    table = g["table"]            # rows: [hi, lo, n, ref8!=exact, ref1!=exact, ref8!=ref1, hip32!=exact, hip64!=exact]
    assert table[:, 6].sum() == 0 and table[:2, 3].sum() == 0 and table[2:, 3].sum() > 0


def test_back_to_back_encodes_with_different_grids_do_not_race(dev):
    """two cogs_vit_encode calls queued on one stream without synchronising in between, different grids (different
    cu_seqlens / row-range tables), both attention modes: each equals the same call made alone -- the tables are
    built by kernels on the stream, nothing is staged in handle-owned host memory"""
    from cogstream_amd.vision import BLOCK_DIAG, REF_EAGER_GLOBAL, VisionEncoder
    from cogstream_amd.weights import VisionConfig, random_vit_state
    cfg = VisionConfig(**VIT)
    enc = VisionEncoder(random_vit_state(cfg, seed=3, std=0.05), cfg, dtype=torch.float32, device=dev)
    g = torch.Generator().manual_seed(2)
    ga, gb = torch.tensor([[3, 4, 6]]), torch.tensor([[2, 6, 4], [5, 2, 2]])
    pa = (torch.rand(72, 588, generator=g) * 2 - 1).to(dev)
    pb = (torch.rand(68, 588, generator=g) * 2 - 1).to(dev)
    ma, mb = torch.tensor([2]), torch.tensor([2, 2])
    for mode in (BLOCK_DIAG, REF_EAGER_GLOBAL):
        torch.cuda.synchronize()
        a = enc(pa, ga, ma, attn_mode=mode).clone()      # the workspace is shared: keep the outputs, not the scratch
        b = enc(pb, gb, mb, attn_mode=mode).clone()
        a2 = enc(pa, ga, ma, attn_mode=mode).clone()
        torch.cuda.synchronize()
        alone_b = enc(pb, gb, mb, attn_mode=mode)
        torch.cuda.synchronize()
        alone_a = enc(pa, ga, ma, attn_mode=mode)
        torch.cuda.synchronize()
        assert torch.equal(a, alone_a) and torch.equal(a2, alone_a) and torch.equal(b, alone_b)


def test_image_modality_vs_reference(dev):
    """tests/golden/image_modality.npz on the HIP path: GPU pre-processing of images (merge size 1) and of an image next
    to a clip (batched_resize) bit-equal to the reference processor's pixel_values; the encoder on an image + a clip in
    ONE call (merge sizes [1, 2]) in both attention modes; the compression mask keeps every token of an image"""
    from cogstream_amd import processing as pr
    from cogstream_amd.preprocess_gpu import preprocess_media_gpu
    from cogstream_amd.vision import BLOCK_DIAG, REF_EAGER_GLOBAL, VisionEncoder
    from cogstream_amd.weights import VisionConfig, random_vit_state
    g = _load("image_modality.npz")
    imgs = [pr.synthetic_clip(1, int(a[0]), int(a[1]), kind=k, clip_idx=int(a[2]))[0] for a, k in zip(g["image_args"], ["noise", "drift"])]
    ca = g["clip_args"]
    clip = pr.synthetic_clip(int(ca[0]), int(ca[1]), int(ca[2]), kind="drift", clip_idx=int(ca[3]))[0]
    for tag, items, merges, kw in (("two", imgs, [1, 1], {}), ("mixed", [clip, imgs[1]], [2, 1], {}),
                                   ("mixed_small", [clip, imgs[1]], [2, 1], {"max_tokens": 40})):
        out = preprocess_media_gpu([torch.from_numpy(np.ascontiguousarray(v)).to(dev) for v in items], merges,
                                   out_dtype=torch.float32, **kw)
        assert out["grid_sizes"].tolist() == g[f"{tag}_grid"].tolist() and out["merge_sizes"].tolist() == merges
        assert torch.equal(out["pixel_values"].cpu(), torch.from_numpy(g[f"{tag}_pixel_values"])), tag
    cfg = VisionConfig(**VIT)
    pix, grid, merge = torch.from_numpy(g["vit_pixel_values"]).to(dev), torch.from_numpy(g["vit_grid"]), torch.from_numpy(g["vit_merge"])
    for dtype, tol in ((torch.float32, 1e-4), (torch.bfloat16, 3e-2)):
        enc = VisionEncoder(random_vit_state(cfg, seed=3, std=0.05), cfg, dtype=dtype, device=dev)
        bd = enc(pix.to(dtype), grid, merge, attn_mode=BLOCK_DIAG)
        eg = enc(pix.to(dtype), grid, merge, attn_mode=REF_EAGER_GLOBAL)
        assert bd.shape == (23, 576)
        assert rel_err(bd.float(), torch.from_numpy(g["vit_block_diag"])) < tol
        assert rel_err(eg.float(), torch.from_numpy(g["vit_eager_global"])) < tol
    model = _tiny_model(dev, torch.float32, 0)
    batched = grid.prod(dim=1).div(merge ** 2).long()
    m = model._get_compression_mask(pix, batched, grid, merge, ["image", "video"], minor_frame_indices=[])
    assert torch.equal(m.cpu(), torch.from_numpy(g["mask_image_video"]))
    # processor -> generate with an image in the conversation (merge size 1 through the whole product path)
    from toy_tokenizer import ToyTokenizer
    proc = pr.CogStreamProcessor(ToyTokenizer(), device=dev, pixel_dtype=torch.float32)
    conv = [{"role": "user", "content": [{"type": "image", "image": imgs[0]}, {"type": "text", "text": "What is this?"}]}]
    inputs = proc(conversation=conv, add_system_prompt=True, add_generation_prompt=True)
    assert inputs["modals"] == ["image"] and inputs["merge_sizes"].tolist() == [1] and inputs["total_image_num"] == 1
    inputs = model.qa_selection(**inputs, mode="FCC")
    ids, _ = model.generate(**inputs, max_new_tokens=4)
    assert ids.shape[0] == 1 and 1 <= ids.shape[1] <= 4
    assert bool(model.last_debug["compression_mask"].all())        # an image keeps all its tokens


def test_kmeans_pp_in_one_call_equals_the_step_by_step_seeding(dev, monkeypatch):
    """cogs_kmeans_pp (the K - 1 k-means++ draws on the device from host-made Exponential(1) rows: torch.multinomial(p, 1)
    on the CPU is argmax(p / q)) against the step-by-step path that calls torch.multinomial itself: the same centres, the
    same final assignment, and BOTH host generators (python random, torch CPU) left at the same position. All-identical
    rows make every probability zero -- the reference's random.randint branch -- and must take the step-by-step path."""
    from cogstream_amd import kmeans as km
    from cogstream_amd import ops
    T, P, D, K = 200, 4, 64, 14
    g = torch.Generator().manual_seed(77)
    cent = torch.randn(K, P * D, generator=g) * 2
    feats = (cent[torch.randint(0, K, (T,), generator=g)] + 0.7 * torch.randn(T, P * D, generator=g)).view(T, P, D)
    ts = torch.arange(T, dtype=torch.float32)
    real_pp = ops.kmeans_pp

    def forced_fallback(*a):                      # pretend a step found all probabilities zero
        idx, flag = real_pp(*a)
        flag.fill_(1)
        return idx, flag

    for dtype in (torch.float32, torch.bfloat16):
        x = feats.to(dev, dtype)
        for seed in (0, 1, 2, 3):
            res = {}
            for path in ("one call", "steps"):
                monkeypatch.setattr(ops, "kmeans_pp", real_pp if path == "one call" else forced_fallback)
                random.seed(seed)
                torch.manual_seed(seed)
                cf, ct, assign = km.kmeans_with_time_min_max(x, ts, K)
                res[path] = (cf.clone(), ct.clone(), assign.clone(), random.random(), float(torch.rand(1)), km.last_stats["kpp_path"])
            a, b = res["one call"], res["steps"]
            assert a[5] == "one call" and b[5].startswith("step by step")
            assert torch.equal(a[2], b[2]) and torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]), (dtype, seed)
            assert a[3] == b[3] and a[4] == b[4], (dtype, seed)               # generator positions
    monkeypatch.setattr(ops, "kmeans_pp", real_pp)
    same = torch.ones(40, 2, 64).to(dev)
    random.seed(5)
    torch.manual_seed(5)
    _, _, assign = km.kmeans_with_time_min_max(same, torch.zeros(40), 6)
    assert km.last_stats["kpp_path"].startswith("step by step") and assign.shape == (40,)


def test_kmeans_margin_statistics_flag_the_planted_near_ties(dev):
    """last_stats["min_rel_margin"] / ["rows_below_1e-3"]: on the near-tie study's inputs (rows planted between two
    centres at relative margins down to 1e-9) the statistic must see margins far below 1e-3 and count those rows; on
    well-separated clusters it must stay above the threshold with a count of zero"""
    from golden.tie_inputs import K, tie_inputs
    from cogstream_amd import kmeans as km
    from cogstream_amd import ops
    inp = tie_inputs()
    x = inp["features"].to(dev)
    T = x.shape[0]
    flat = x.view(T, -1)
    ws = ops.kmeans_workspace(T, flat.shape[1], K, dev)
    centres = flat[:K].clone().contiguous()
    cts = torch.zeros(K, device=dev)
    assign = torch.empty(T, dtype=torch.int64, device=dev)
    it, _, _ = ops.kmeans_lloyd(flat, torch.zeros(T, device=dev), centres, cts, assign, 2.0, 1, 1e-4, [0] * 8, ws)
    m, n = ops.kmeans_margins(T, flat.shape[1], K, ws)
    margin = torch.from_numpy(_load("kmeans_ties.npz")["margin"])[K:]           # per planted row, fp64, of the feature distance
    # fp32 resolves the planted margins down to ~1e-6; rows within 10 % of the threshold may fall on either side
    assert it == 1 and 0 <= m < 1e-5 and int((margin < 9e-4).sum()) <= n <= int((margin < 1.1e-3).sum())
    g = torch.Generator().manual_seed(1)
    cent = torch.randn(6, 256, generator=g) * 10
    sep = (cent[torch.arange(120) % 6] + 0.1 * torch.randn(120, 256, generator=g)).view(120, 1, 256).to(dev)
    random.seed(0)
    torch.manual_seed(0)
    km.kmeans_with_time_min_max(sep, torch.zeros(120), 6)
    assert km.last_stats["rows_below_1e-3"] == 0 and km.last_stats["min_rel_margin"] > 1e-3


def test_maybe_truncate_visual_tokens_vs_reference(dev):
    """_maybe_truncate_visual_tokens (cogreasoner_chat.py:349-381; tests/golden/truncate.npz made by the reference's own
    method): a packed row whose first sample kept 5 of its 8 <image> placeholders loses the 3 surplus visual tokens and
    mask entries; without position_ids, or when the counts already agree, the inputs come back untouched"""
    g = _load("truncate.npz")
    model = _tiny_model(dev, torch.float32, 0)
    model.config.image_token_index = 258
    mm, mask = torch.from_numpy(g["mm"]).to(dev), torch.from_numpy(g["mask"]).to(dev)
    batched, ids, pos = torch.from_numpy(g["batched"]), torch.from_numpy(g["input_ids"]), torch.from_numpy(g["position_ids"])
    out_mm, out_mask = model._maybe_truncate_visual_tokens(mm, mask, batched, ["video", "image"], ids, pos)
    assert torch.equal(out_mm.cpu(), torch.from_numpy(g["out_mm"])) and torch.equal(out_mask.cpu(), torch.from_numpy(g["out_mask"]))
    a, b = model._maybe_truncate_visual_tokens(mm, mask, batched, ["video", "image"], ids, None)
    assert a is mm and b is mask
    a, b = model._maybe_truncate_visual_tokens(mm, mask, batched, ["video", "image"], torch.from_numpy(g["input_ids_full"]), torch.arange(16))
    assert a is mm and b is mask
