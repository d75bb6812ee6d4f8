"""Pin the oracle (CPU restatement) against fixtures produced by RUNNING THE REFERENCE
(tests/golden/make_golden.py, build container only). fp32 on both sides: tolerances are summation-order
noise; integer outputs (assignments, masks, ids) must be bit-exact."""
import json
import os
import random

import numpy as np
import pytest
import torch

from conftest import rel_err

G = os.path.join(os.path.dirname(__file__), "golden")
VIT = dict(hidden_size=576, intermediate_size=200, num_hidden_layers=2, num_attention_heads=8)


def _load(name):
    return {k: v for k, v in np.load(os.path.join(G, name), allow_pickle=False).items()}


def test_vit_oracle_matches_reference():
    from cogstream_amd.weights import VisionConfig, random_proj_state, random_vit_state
    from oracle import vision as ov
    g = _load("vit_tiny.npz")
    cfg = VisionConfig(**VIT)
    st = random_vit_state(cfg, seed=3, std=0.05)
    assert abs(float(sum(v.double().abs().sum() for v in st.values())) - float(g["vit_checksum"])) < 1e-6
    pix, grid, merge = torch.from_numpy(g["pixel_values"]), torch.from_numpy(g["grid_sizes"]), torch.from_numpy(g["merge_sizes"])
    kw = dict(heads=cfg.num_attention_heads, layers=cfg.num_hidden_layers)
    eg = ov.encode(st, pix, grid, merge, mode=ov.REF_EAGER_GLOBAL, **kw)
    bd = ov.encode(st, pix, grid, merge, mode=ov.BLOCK_DIAG, **kw)
    assert rel_err(eg, torch.from_numpy(g["eager_global"])) < 2e-5
    assert rel_err(bd, torch.from_numpy(g["block_diag"])) < 2e-5
    # the two attention semantics really differ (SURVEY.md headline fact 2)
    assert rel_err(eg, bd) > 1e-3
    pst = random_proj_state(cfg.hidden_size, 256, seed=1, std=0.05)
    assert rel_err(ov.project(pst, bd), torch.from_numpy(g["projected"])) < 2e-5


def test_rope_and_patch_order_contracts():
    """SURVEY.md appendix B1/B2 (probed on the reference)"""
    from oracle import vision as ov
    ids = ov.rot_pos_ids(torch.tensor([[1, 4, 6]]), torch.tensor([2]))
    assert ids[:8].tolist() == [[0, 0], [0, 1], [1, 0], [1, 1], [0, 2], [0, 3], [1, 2], [1, 3]]
    order = ov.patchify_order(2, 4, 6, 2)
    assert order[:5].tolist() == [[0, 0, 0], [0, 0, 1], [0, 1, 0], [0, 1, 1], [0, 0, 2]]
    f = ov.rotary_freqs(torch.tensor([[1, 4, 6]]), torch.tensor([2]), 72)
    assert f.shape == (24, 36)
    inv = 1.0 / (10000.0 ** (torch.arange(0, 36, 2).float() / 36))
    assert torch.allclose(inv[:4], torch.tensor([1.0, 0.59948, 0.35938, 0.21544]), atol=1e-5)
    assert torch.allclose(f[3], torch.cat([1 * inv, 1 * inv]))  # row 3 = (h=1, w=1)


@pytest.mark.parametrize("ci", [0, 1, 2, 3])
def test_kmeans_oracle_matches_reference(ci):
    from oracle import kmeans as ok
    g = _load("kmeans.npz")
    feats = torch.from_numpy(g[f"c{ci}_features"])
    if int(g[f"c{ci}_is_bf16"]):
        feats = feats.bfloat16()
    ts, K, seed = torch.from_numpy(g[f"c{ci}_ts"]), int(g[f"c{ci}_K"]), int(g[f"c{ci}_seed"])
    random.seed(seed)
    torch.manual_seed(seed)
    cf, ct, assign = ok.kmeans_with_time_min_max(feats, ts, K)
    assert rel_err(cf.float(), torch.from_numpy(g[f"c{ci}_centres"])) < 1e-5
    assert rel_err(ct, torch.from_numpy(g[f"c{ci}_centre_ts"])) < 1e-6
    if f"c{ci}_assign" in g:
        assert torch.equal(assign, torch.from_numpy(g[f"c{ci}_assign"]))
        sel = ok.select_additional_frames(feats, cf, assign, 2)
        assert torch.equal(torch.cat(sel).sort().values, torch.from_numpy(g[f"c{ci}_extra"]))
        assert [len(s) for s in sel] == g[f"c{ci}_extra_counts"].tolist()
    else:
        assert assign is None


@pytest.mark.parametrize("ci", [0, 1])
def test_kmeans_oracle_reseed_branch_matches_reference(ci):
    """tests/golden/kmeans_reseed.npz: the reference on degenerate inputs whose duplicate centres leave clusters empty
    (3 reseeds once / 7 reseeds in each of 30 iterations) -- assignments, centres and the generator position after the call"""
    import random
    from oracle import kmeans as ok
    g = _load("kmeans_reseed.npz")
    random.seed(int(g[f"c{ci}_seed"]))
    torch.manual_seed(int(g[f"c{ci}_seed"]))
    cf, ct, assign = ok.kmeans_with_time_min_max(torch.from_numpy(g[f"c{ci}_features"]), torch.from_numpy(g[f"c{ci}_ts"]), int(g[f"c{ci}_K"]))
    assert random.random() == float(g[f"c{ci}_next_random"])
    assert torch.equal(assign, torch.from_numpy(g[f"c{ci}_assign"]))
    assert rel_err(cf, torch.from_numpy(g[f"c{ci}_centres"])) < 1e-5 and rel_err(ct + 1, torch.from_numpy(g[f"c{ci}_centre_ts"]) + 1) < 1e-6


def test_compress_oracle_matches_reference():
    from oracle import compress as oc
    g = _load("compress.npz")
    pix = torch.from_numpy(g["pixel_values"])
    grid, merge = torch.from_numpy(g["grid_sizes"]), torch.from_numpy(g["merge_sizes"])
    for tag, px in (("f32", pix), ("bf16", pix.bfloat16())):
        assert torch.equal(oc.compression_mask(px, grid, merge, ["video"]), torch.from_numpy(g[f"mask_{tag}"]))
        assert torch.equal(oc.compression_mask(px, grid, merge, ["video"], minor_frame_indices=[2, 5]),
                           torch.from_numpy(g[f"mask_minor_{tag}"]))
    m = torch.from_numpy(g["mask_f32"])
    assert 0 < int(m.sum()) < m.numel()
    mm = torch.from_numpy(g["mm"])
    ev = oc.compress_unimportant_events(mm, 6, [1, 4])
    assert torch.equal(ev, torch.from_numpy(g["event_pooled"]))
    mm2, ids2, _ = oc.compress_visual_tokens(torch.from_numpy(g["mask_minor_f32"]), ev, torch.from_numpy(g["input_ids"]),
                                             None, 258)
    assert torch.equal(ids2, torch.from_numpy(g["ids_compressed"])) and torch.equal(mm2, torch.from_numpy(g["mm_compressed"]))


def test_host_text_logic_matches_reference():
    """prompt surgery / prompts are host logic of the product (cogstream_amd.chat / qaselect): same strings"""
    from cogstream_amd.chat import create_visual_summary_prompt, parse_selection, process_input_ids
    from cogstream_amd.qaselect import format_example
    from toy_tokenizer import ToyTokenizer
    tok = ToyTokenizer()
    t = json.load(open(os.path.join(G, "text.json")))
    for c in t["cases"]:
        vis, sel = parse_selection(c["selection"])
        hq = [t["hist_qs"][i] for i in sel if i < len(t["hist_qs"])]
        ha = [t["hist_as"][i] for i in sel if i < len(t["hist_qs"])]
        out = process_input_ids(t["original_text"], vis, hq, ha, t["current_question"])
        assert vis == c["if_visual"], c["selection"]
        assert out == c["prompt"], c["selection"]
        assert tok.encode(out) == c["ids"]
    assert create_visual_summary_prompt(6, torch.tensor([0.0, 1.04, 12.5])) == t["summary_prompt"]
    ex = {"current_Q": t["current_question"], "hist_Qs": t["hist_qs"], "hist_As": t["hist_as"]}
    assert format_example(ex) == t["qa_prompt"]
    assert format_example(ex, include_demo=False) == t["qa_prompt_nodemo"]


def test_preprocessing_matches_reference_processor_bit_exact():
    """host pre-processing against the reference's Videollama3ImageProcessor output (tests/golden/make_golden.py)"""
    from cogstream_amd import processing as pr
    g = np.load(os.path.join(G, "preprocess.npz"))
    clips = [pr.synthetic_clip(int(a[0]), int(a[1]), int(a[2]), kind=k, clip_idx=int(a[3]))[0]
             for a, k in zip(g["clip_args"], ["drift", "noise"])]
    out = pr.preprocess_videos(clips)
    assert out["grid_sizes"].tolist() == g["grid_sizes"].tolist()
    assert out["merge_sizes"].tolist() == g["merge_sizes"].tolist()
    assert np.array_equal(out["pixel_values"], g["pixel_values"])
    ramp = pr.preprocess_videos([g["ramp"]])
    assert ramp["grid_sizes"].tolist() == g["ramp_grid"].tolist()
    assert np.array_equal(ramp["pixel_values"], g["ramp_pixel_values"])
    # the pure-numpy resample restatement (what the HIP kernels implement) gives the same image as PIL
    fr = clips[0][0]
    th, tw = int(g["grid_sizes"][0][1]) * 14, int(g["grid_sizes"][0][2]) * 14
    assert np.array_equal(pr.resize_bicubic_exact(fr, (th, tw)), pr._resize_bicubic(fr, (th, tw)))


def test_merge_lora_equals_unmerged_branch_fp32():
    """weights.merge_lora (what the product loads) against the oracle's unmerged peft branch, on the CPU in fp32"""
    from oracle import qwen2 as oq
    from cogstream_amd.weights import LlmConfig, merge_lora, random_llm_state, random_lora_state
    cfg = LlmConfig(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4,
                    num_key_value_heads=2, vocab_size=96)
    st = random_llm_state(cfg, seed=3, std=0.08)
    lora = random_lora_state(cfg, seed=4, std=0.08)
    kw = dict(heads=4, kv_heads=2, layers=2)
    torch.manual_seed(0)
    emb = torch.randn(19, 64)
    merged, none = merge_lora(st, None, lora, cfg, lora_alpha=16.0)
    assert none is None and merged["embed_tokens.weight"] is st["embed_tokens.weight"]      # untouched tensors are shared
    a = oq.forward(merged, emb, **kw)[0]
    b = oq.forward(st, emb, lora={k.replace("base_model.model.model.", ""): v for k, v in lora.items()},
                   lora_scaling=2.0, **kw)[0]
    assert rel_err(a, b) < 1e-5
    assert rel_err(a, oq.forward(st, emb, **kw)[0]) > 0.05


def _lora_case():
    import json as _json
    from cogstream_amd.weights import (LlmConfig, VisionConfig, random_llm_state, random_lora_state, random_proj_state,
                                       random_vit_state)
    g = _load("lora.npz")
    lcfg = LlmConfig(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2,
                     num_key_value_heads=1, vocab_size=512, image_token_index=258, eos_token_id=257)
    lst = random_llm_state(lcfg, seed=7, std=0.05)
    pst = random_proj_state(576, 256, seed=1, std=0.05)
    assert abs(float(sum(v.double().abs().sum() for v in lst.values())) - float(g["llm_checksum"])) < 1e-6
    loras = {}
    for a in ("full_module", "language_module"):
        lo = random_lora_state(lcfg, seed=int(g[f"{a}_seed"]), r=int(g["r"]), proj_dims=(576, 256))
        assert abs(float(sum(v.double().abs().sum() for v in lo.values())) - float(g[f"{a}_checksum"])) < 1e-6
        loras[a] = lo
    targets = _json.loads(str(g["targets"]))
    return g, lcfg, lst, pst, loras, targets


def test_lora_branch_matches_the_reference_model_with_peft_style_modules():
    """tests/golden/lora.npz: the REFERENCE's model object with every linear of the reference's target-module list
    (train/second_stage_training.py:241-254) wrapped by an independently written module computing peft's published
    eval-mode forward, base(x) + lora_B(lora_A(x)) * (lora_alpha / r), r = 8, alpha = 16 (:257-264). Pins (a) the
    restated unmerged branch of oracle/qwen2.py::lora_linear, (b) weights.merge_lora (what the product loads) and (c)
    the adapter key layout / target list. peft's own code could not be run in this image (not installed)."""
    from oracle import qwen2 as oq
    from oracle import vision as ov
    from cogstream_amd.weights import LORA_LLM_TARGETS, merge_lora
    g, lcfg, lst, pst, loras, targets = _lora_case()
    assert sorted(targets) == sorted([f"model.layers.{i}.{t}" for i in range(2) for t in LORA_LLM_TARGETS] +
                                     ["model.mm_projector.readout.0", "model.mm_projector.readout.2"])
    kw = dict(heads=2, kv_heads=1, layers=2)
    emb, vis = torch.from_numpy(g["embeds"]), torch.from_numpy(g["vis"])
    scaling = float(g["lora_alpha"]) / int(g["r"])
    base = oq.forward(lst, emb, **kw)[0]
    assert rel_err(base, torch.from_numpy(g["base_hidden"])) < 2e-5
    assert rel_err(ov.project(pst, vis), torch.from_numpy(g["base_projected"])) < 2e-5
    for a, lo in loras.items():
        strip = {k.replace("base_model.model.model.", ""): v for k, v in lo.items()}
        hid = oq.forward(lst, emb, lora=strip, lora_scaling=scaling, **kw)[0]                  # (a) unmerged branch
        assert rel_err(hid, torch.from_numpy(g[f"{a}_hidden"])) < 2e-5
        assert rel_err(oq.logits(lst, hid[-1]), torch.from_numpy(g[f"{a}_logits"])) < 2e-5
        ml, mp = merge_lora(lst, pst, lo, lcfg, lora_alpha=float(g["lora_alpha"]))             # (b) merged weights
        assert rel_err(oq.forward(ml, emb, **kw)[0], torch.from_numpy(g[f"{a}_hidden"])) < 2e-5
        assert rel_err(ov.project(mp, vis), torch.from_numpy(g[f"{a}_projected"])) < 2e-5
        assert rel_err(torch.from_numpy(g[f"{a}_hidden"]), torch.from_numpy(g["base_hidden"])) > 0.05   # the adapter acts


def _qwen2_case(g, tag):
    import json as _json
    from cogstream_amd.weights import LlmConfig, random_llm_state
    llm = _json.loads(str(g[f"{tag}_cfg"]))
    lcfg = LlmConfig(**llm)
    lst = random_llm_state(lcfg, seed=7, std=0.05)
    assert abs(float(sum(v.double().abs().sum() for v in lst.values())) - float(g[f"{tag}_llm_checksum"])) < 1e-6
    return lcfg, lst


@pytest.mark.parametrize("tag", ["t", "g"])
def test_qwen2_oracle_matches_reference(tag):
    """oracle/qwen2.py against the reference's own model object (tests/golden/qwen2.npz: last_hidden_state,
    sequence mean, prefill logits and three cached decode steps; model/cogreasoner_chat.py:312-323,802-807)."""
    from oracle import qwen2 as oq
    g = _load("qwen2.npz")
    lcfg, lst = _qwen2_case(g, tag)
    kw = dict(heads=lcfg.num_attention_heads, kv_heads=lcfg.num_key_value_heads, layers=lcfg.num_hidden_layers,
              eps=lcfg.rms_norm_eps, theta=lcfg.rope_theta)
    emb = torch.from_numpy(g[f"{tag}_embeds"])
    hid, kv = oq.forward(lst, emb, **kw)
    ref_lg = torch.from_numpy(g[f"{tag}_logits_f32"])
    assert rel_err(hid, torch.from_numpy(g[f"{tag}_hidden_f32"])) < 2e-5
    assert rel_err(hid.mean(0), torch.from_numpy(g[f"{tag}_pooled_f32"])) < 2e-5
    assert rel_err(oq.logits(lst, hid[-1]), ref_lg[0]) < 2e-5
    for i, tok in enumerate(g[f"{tag}_step_tokens"].tolist()):
        assert int(ref_lg[i].argmax()) == tok                          # the steps ARE the greedy continuation
        h1, kv = oq.forward(lst, lst["embed_tokens.weight"][tok][None], past=kv, **kw)
        assert rel_err(oq.logits(lst, h1[-1]), ref_lg[i + 1]) < 2e-5
    # the oracle evaluated in bf16 follows the reference evaluated in bf16 (same op order, same roundings):
    # not bit-equal (CPU bf16 matmul blocking differs with shapes) but far inside the bf16-vs-fp32 distance
    lst16 = {k: v.bfloat16() for k, v in lst.items()}
    h16, _ = oq.forward(lst16, emb.bfloat16(), **kw)
    ref16, ref32 = torch.from_numpy(g[f"{tag}_hidden_bf16"]), torch.from_numpy(g[f"{tag}_hidden_f32"])
    assert rel_err(h16.float(), ref16) <= 1.0 * rel_err(ref16, ref32)


def test_image_modality_preprocessing_encoder_and_mask_vs_reference():
    """tests/golden/image_modality.npz (made by the reference's Videollama3ImageProcessor, its encoder and
    _get_compression_mask): images use merge size 1; two images share simple_batched_resize, an image next to a clip goes
    through batched_resize (over and under the token budget); the oracle encoder handles merge sizes [1, 2] in one
    call; an image keeps every token in the compression mask (cogreasoner_chat.py:400-403)"""
    from cogstream_amd import processing as pr
    from cogstream_amd.weights import VisionConfig, random_vit_state
    from oracle import compress as oc
    from oracle import vision as ov
    g = np.load(os.path.join(G, "image_modality.npz"))
    imgs = [pr.synthetic_clip(1, int(a[0]), int(a[1]), kind=k, clip_idx=int(a[2]))[0] for a, k in zip(g["image_args"], ["noise", "drift"])]
    ca = g["clip_args"]
    clip = pr.synthetic_clip(int(ca[0]), int(ca[1]), int(ca[2]), kind="drift", clip_idx=int(ca[3]))[0]
    for tag, items, merges, kw in (("two", imgs, [1, 1], {}), ("mixed", [clip, imgs[1]], [2, 1], {}),
                                   ("mixed_small", [clip, imgs[1]], [2, 1], {"max_tokens": 40})):
        out = pr.preprocess_media(items, merges, **kw)
        assert out["grid_sizes"].tolist() == g[f"{tag}_grid"].tolist(), tag
        assert np.array_equal(out["pixel_values"], g[f"{tag}_pixel_values"]), tag
    cfg = VisionConfig(hidden_size=576, intermediate_size=200, num_hidden_layers=2, num_attention_heads=8)
    st = random_vit_state(cfg, seed=3, std=0.05)
    assert abs(float(sum(v.double().abs().sum() for v in st.values())) - float(g["vit_checksum"])) < 1e-6
    pix, grid, merge = torch.from_numpy(g["vit_pixel_values"]), torch.from_numpy(g["vit_grid"]), torch.from_numpy(g["vit_merge"])
    for mode, key in ((0, "vit_block_diag"), (1, "vit_eager_global")):
        tok = ov.encode(st, pix, grid, merge, heads=8, layers=2, mode=mode)
        assert tok.shape == (15 + 8, 576) and rel_err(tok, torch.from_numpy(g[key])) < 2e-5
    batched = grid.prod(dim=1).div(merge ** 2).long()
    mask = oc.compression_mask(pix, grid, merge, ["image", "video"], minor_frame_indices=[])
    assert torch.equal(mask, torch.from_numpy(g["mask_image_video"])) and bool(mask[:15].all()) and int(batched[0]) == 15


def test_processor_call_surface_matches_the_reference():
    """Videollama3Qwen2Processor.__call__(text=None, conversation=None, images=None, return_labels=False, **kwargs)
    (processing_cogreasoner.py:732-744): argument order, the two error messages, chat-template defaults (no system
    turn, no generation prompt unless asked), the plain-text path, image items (merge 1, 'Time x.xs: <image>\\n'),
    media handed over through images= (the path that dies with a NameError in the reference, :639-641)"""
    import inspect
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from toy_tokenizer import ToyTokenizer
    from cogstream_amd import processing as pr
    from cogstream_amd.processing_cogreasoner import Videollama3Qwen2Processor
    proc = Videollama3Qwen2Processor(ToyTokenizer())
    assert list(inspect.signature(proc.__call__).parameters)[:4] == ["text", "conversation", "images", "return_labels"]
    with pytest.raises(ValueError, match="You cannot provide 'message' with 'text'."):
        proc(text="x", conversation=[{"role": "user", "content": "q"}])
    with pytest.raises(ValueError, match="You must provide 'text' or 'message'."):
        proc()
    with pytest.raises(ValueError, match="return_labels is not supported for plain text processing."):
        proc(text="x", return_labels=True)
    with pytest.raises(NotImplementedError):
        proc(conversation=[{"role": "user", "content": "q"}], return_labels=True)
    # chat-template defaults of the reference: nothing added unless asked for
    out = proc(conversation=[{"role": "user", "content": "q?"}])
    assert out["original_text"] == "<|im_start|>user\nq?<|im_end|>\n" and out["modals"] == [] and "pixel_values" not in out
    out = proc(conversation=[{"role": "user", "content": "q?"}], add_system_prompt=True, add_generation_prompt=True)
    assert out["original_text"].startswith("<|im_start|>system\n" + pr.DEFAULT_SYSTEM) and out["original_text"].endswith("<|im_start|>assistant\n")
    # image + clip in one conversation
    img = pr.synthetic_clip(1, 75, 110, kind="noise", clip_idx=11)[0][0]
    clip, ts = pr.synthetic_clip(3, 60, 100, kind="drift", clip_idx=13)
    conv = [{"role": "user", "content": [{"type": "image", "image": img, "timestamp": 2.04}, {"type": "video", "video": clip, "timestamps": ts},
                                         {"type": "text", "text": "What changed?"}]}]
    out = proc(conversation=conv, add_generation_prompt=True)
    assert out["modals"] == ["image", "video"] and out["merge_sizes"].tolist() == [1, 2] and out["total_image_num"] == 4
    gh, gw = out["grid_sizes"][0, 1:].tolist()
    per = [gh * gw] + [int(g[1] * g[2]) // 4 for g in out["grid_sizes"][1:].tolist() for _ in range(3)]
    assert out["original_text"].startswith("<|im_start|>user\nTime 2.0s: " + "<image>" * per[0] + "\nTime 0.0s:" + "<image>" * per[1] + ",")
    assert int((out["input_ids"] == 258).sum()) == sum(per) and out["all_timestamps"] == [2.04, 0.0, 1.0, 2.0]
    assert out["pixel_values"].shape[0] == gh * gw + 3 * int(out["grid_sizes"][1, 1] * out["grid_sizes"][1, 2])
    # the same media through images= (named and bare), conversation items then only describe them
    conv2 = [{"role": "user", "content": [{"type": "image", "timestamp": 2.04}, {"type": "video", "num_frames": 3, "timestamps": ts},
                                          {"type": "text", "text": "What changed?"}]}]
    for given in ([("image", img), ("video", clip)], [img, clip]):
        o2 = proc(conversation=conv2, images=given, add_generation_prompt=True)
        assert torch.equal(o2["input_ids"], out["input_ids"]) and torch.equal(o2["pixel_values"], out["pixel_values"])
        assert o2["video_keys"] == out["video_keys"]
    # plain text + images (process_text expands the placeholders): what _process_plain means to return
    plain = proc(text="look: <image> and <image>", images=[img, img])
    assert set(plain) == {"input_ids", "attention_mask", "pixel_values", "grid_sizes", "merge_sizes", "modals"}
    assert int((plain["input_ids"] == 258).sum()) == 2 * gh * gw and plain["modals"] == ["image", "image"]
    assert set(proc(text="no media")) == {"input_ids", "attention_mask"}
