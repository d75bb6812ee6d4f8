#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE (imported from
/root/reference, CPU, fp32, eager) on small seeded inputs. Runs only in the build container -- the
reference does not exist on the GPU box; the fixtures (inputs + expected outputs, data only) are committed.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Weights are not stored: both sides regenerate them with cogstream_amd.weights.random_*_state(seed) (torch CPU
generator, same image on both boxes); each fixture stores a weight checksum so RNG drift is detected."""
import copy
import json
import os
import random
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True

from model import cogreasoner_chat as ref_chat  # noqa: E402
from model.configuration_videollama3 import Videollama3Qwen2Config  # noqa: E402
from model.configuration_videollama3_encoder import Videollama3VisionEncoderConfig  # noqa: E402
from model.kmeans_with_time import kmeans_with_time_min_max as ref_kmeans  # noqa: E402
from model.modeling_videollama3_encoder import Videollama3VisionEncoderModel  # noqa: E402
from model.qaselect_module_predict import format_example as ref_format_example  # noqa: E402

from cogstream_amd.weights import (LlmConfig, VisionConfig, random_llm_state, random_proj_state,  # noqa: E402
                                   random_vit_state)
from toy_tokenizer import IM_END, IMAGE, VOCAB, ToyTokenizer  # noqa: E402
from golden.inputs import FORCED_COSINE, e2e_inputs  # noqa: E402

VIT = dict(hidden_size=576, intermediate_size=200, num_hidden_layers=2, num_attention_heads=8)
LLM = dict(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2,
           num_key_value_heads=1, vocab_size=VOCAB, image_token_index=IMAGE, eos_token_id=IM_END)


def checksum(state):
    return float(sum(v.double().abs().sum() for v in state.values()))


def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = v
    np.savez_compressed(os.path.join(HERE, name), **out)
    print("wrote", name, {k: getattr(v, "shape", None) for k, v in out.items()})


def ref_vit(state, cfg):
    c = Videollama3VisionEncoderConfig(hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size,
                                       num_hidden_layers=cfg.num_hidden_layers,
                                       num_attention_heads=cfg.num_attention_heads, patch_size=14)
    c._attn_implementation = "eager"
    m = Videollama3VisionEncoderModel(c).eval()
    missing = m.load_state_dict(state, strict=True)
    return m


def golden_vit():
    cfg = VisionConfig(**VIT)
    st = random_vit_state(cfg, seed=3, std=0.05)
    m = ref_vit(st, cfg)
    g = torch.Generator().manual_seed(21)
    grid = torch.tensor([[3, 4, 6], [2, 2, 4]])
    merge = torch.tensor([2, 2])
    n = int(grid.prod(1).sum())
    pix = torch.rand(n, 588, generator=g) * 2 - 1
    with torch.no_grad():
        eager_global = m(pix, grid, merge)                     # reference CPU path: global attention + 1.0 bias
        # block-diagonal (= the flash path the reference ships with): one frame per call
        outs, row = [], 0
        for (t, h, w), ms in zip(grid.tolist(), merge.tolist()):
            for f in range(t):
                outs.append(m(pix[row:row + h * w], torch.tensor([[1, h, w]]), torch.tensor([ms])))
                row += h * w
        block_diag = torch.cat(outs, 0)
    pst = random_proj_state(cfg.hidden_size, LLM["hidden_size"], seed=1, std=0.05)
    cfg_ns = types.SimpleNamespace(vision_encoder_config=types.SimpleNamespace(hidden_size=cfg.hidden_size),
                                   hidden_size=LLM["hidden_size"])
    proj = ref_chat.MlpGeluProjector(cfg_ns, "mlp2x_gelu").eval()
    proj.load_state_dict(pst, strict=True)
    with torch.no_grad():
        projected = proj(block_diag)
    save("vit_tiny.npz", pixel_values=pix, grid_sizes=grid, merge_sizes=merge, eager_global=eager_global,
         block_diag=block_diag, projected=projected, vit_checksum=np.float64(checksum(st)),
         proj_checksum=np.float64(checksum(pst)))


def golden_preprocess():
    """Videollama3ImageProcessor.preprocess (the reference's own class) on two small seeded clips of different
    sizes; stores the reference's pixel_values / grid_sizes, plus the per-byte value table it implies."""
    # transformers 5.x (this image) moved the `VideoInput` type alias from image_utils to video_utils; the
    # reference (pinned to 4.46.3) imports it from the old place and uses it in annotations only.
    import transformers.image_utils as _iu
    import transformers.video_utils as _vu
    if not hasattr(_iu, "VideoInput"):
        _iu.VideoInput = _vu.VideoInput
    from model.image_processing_videollama3 import Videollama3ImageProcessor
    from cogstream_amd.processing import synthetic_clip
    with open("/root/reference/model/preprocessor_config.json") as f:       # the shipped processor settings
        pc = json.load(f)
    proc = Videollama3ImageProcessor(**{k: pc[k] for k in ("do_resize", "resample", "do_rescale", "rescale_factor",
                                                           "do_normalize", "image_mean", "image_std", "do_convert_rgb",
                                                           "min_tokens", "max_tokens", "patch_size")})
    clips = [synthetic_clip(2, 60, 100, kind="drift", clip_idx=5)[0], synthetic_clip(1, 90, 70, kind="noise", clip_idx=6)[0]]
    out = proc.preprocess(images=[[f for f in c] for c in clips], merge_size=2, return_tensors="np")
    ramp = np.arange(256, dtype=np.uint8).reshape(1, 16, 16, 1).repeat(3, axis=3)
    ramp = np.tile(ramp, (1, 2, 2, 1))[:, :28, :28]                 # one 28x28 frame, no resize needed
    lut_out = proc.preprocess(images=[[ramp[0]]], merge_size=2, return_tensors="np")
    save("preprocess.npz", clip_args=np.array([[2, 60, 100, 5], [1, 90, 70, 6]]),
         pixel_values=out["pixel_values"].astype(np.float32), grid_sizes=out["grid_sizes"],
         merge_sizes=out["merge_sizes"], ramp=ramp, ramp_pixel_values=lut_out["pixel_values"].astype(np.float32),
         ramp_grid=lut_out["grid_sizes"])


def golden_image_modality():
    """the IMAGE modality (merge_size 1) through the reference's own objects:
      * Videollama3ImageProcessor.preprocess on two images (merge 1 for all: simple_batched_resize with factor 14,
        image_processing_videollama3.py:424-431) and on [video clip, image] with merge sizes [2, 1] (batched_resize,
        :432-439): pixel_values, grid_sizes;
      * the reference encoder on an image + a clip in one call, merge sizes [1, 2], one frame per call (the shipped
        flash semantics) and its eager global form;
      * _get_compression_mask's image branch (cogreasoner_chat.py:400-403): all True for an image."""
    import transformers.image_utils as _iu
    import transformers.video_utils as _vu
    if not hasattr(_iu, "VideoInput"):
        _iu.VideoInput = _vu.VideoInput
    from model.image_processing_videollama3 import Videollama3ImageProcessor
    from cogstream_amd.processing import synthetic_clip
    with open("/root/reference/model/preprocessor_config.json") as f:
        pc = json.load(f)
    proc = Videollama3ImageProcessor(**{k: pc[k] for k in ("do_resize", "resample", "do_rescale", "rescale_factor",
                                                           "do_normalize", "image_mean", "image_std", "do_convert_rgb",
                                                           "min_tokens", "max_tokens", "patch_size")})
    img_a = synthetic_clip(1, 75, 110, kind="noise", clip_idx=11)[0][0]
    img_b = synthetic_clip(1, 130, 64, kind="drift", clip_idx=12)[0][0]
    clip = synthetic_clip(3, 60, 100, kind="drift", clip_idx=13)[0]
    two = proc.preprocess(images=[img_a, img_b], merge_size=[1, 1], return_tensors="np")
    mixed = proc.preprocess(images=[[f for f in clip], img_b], merge_size=[2, 1], return_tensors="np")
    # a budget small enough to shrink the mixed batch (batched_resize's total_tokens > max_tokens branch)
    proc_small = Videollama3ImageProcessor(**{**{k: pc[k] for k in ("do_resize", "resample", "do_rescale", "rescale_factor",
                                                                    "do_normalize", "image_mean", "image_std", "do_convert_rgb",
                                                                    "min_tokens", "patch_size")}, "max_tokens": 40})
    mixed_small = proc_small.preprocess(images=[[f for f in clip], img_b], merge_size=[2, 1], return_tensors="np")
    out = {"image_args": np.array([[75, 110, 11], [130, 64, 12]]), "clip_args": np.array([3, 60, 100, 13]),
           "two_pixel_values": two["pixel_values"].astype(np.float32), "two_grid": two["grid_sizes"], "two_merge": two["merge_sizes"],
           "mixed_pixel_values": mixed["pixel_values"].astype(np.float32), "mixed_grid": mixed["grid_sizes"],
           "mixed_merge": mixed["merge_sizes"], "mixed_small_pixel_values": mixed_small["pixel_values"].astype(np.float32),
           "mixed_small_grid": mixed_small["grid_sizes"]}
    # encoder: one image (merge 1) + one 2-frame clip (merge 2)
    cfg = VisionConfig(**VIT)
    st = random_vit_state(cfg, seed=3, std=0.05)
    m = ref_vit(st, cfg)
    g = torch.Generator().manual_seed(23)
    grid = torch.tensor([[1, 3, 5], [2, 4, 4]])
    merge = torch.tensor([1, 2])
    pix = torch.rand(int(grid.prod(1).sum()), 588, generator=g) * 2 - 1
    with torch.no_grad():
        eager_global = m(pix, grid, merge)
        outs, row = [], 0
        for (t, h, w), ms in zip(grid.tolist(), merge.tolist()):
            for _ in range(t):
                outs.append(m(pix[row:row + h * w], torch.tensor([[1, h, w]]), torch.tensor([ms])))
                row += h * w
        block_diag = torch.cat(outs, 0)
    out.update(vit_pixel_values=pix, vit_grid=grid, vit_merge=merge, vit_eager_global=eager_global, vit_block_diag=block_diag,
               vit_checksum=np.float64(checksum(st)))
    # compression mask with an image among the inputs
    ns = types.SimpleNamespace()
    batched = grid.prod(dim=1).div(merge ** 2).long()
    mask = ref_chat.Videollama3MetaForCausalLM._get_compression_mask(ns, pix, batched, grid, merge, ["image", "video"],
                                                                     minor_frame_indices=[])
    out["mask_image_video"] = mask
    save("image_modality.npz", **out)


def golden_truncate():
    """_maybe_truncate_visual_tokens (cogreasoner_chat.py:349-381), the reference's own method on a packed two-sample
    row: position_ids restart at the second sample, the first sample's text was cut so that only 5 of its 8 visual
    placeholders survive -- the surplus visual tokens (and mask entries) must go. Also the two identity cases
    (position_ids None; counts already equal)."""
    ns = types.SimpleNamespace(config=types.SimpleNamespace(image_token_index=IMAGE))
    fn = ref_chat.Videollama3MetaForCausalLM._maybe_truncate_visual_tokens
    g = torch.Generator().manual_seed(9)
    mm = torch.randn(8 + 6, 16, generator=g)
    mask = torch.rand(14, generator=g) > 0.3
    batched = torch.tensor([8, 6])
    modals = ["video", "image"]
    ids = torch.tensor([1, 2] + [IMAGE] * 5 + [3, 4] + [7] + [IMAGE] * 6 + [9])
    pos = torch.tensor(list(range(9)) + list(range(8)))
    out_mm, out_mask = fn(ns, mm, mask, batched, modals, ids, pos)
    same_mm, same_mask = fn(ns, mm, mask, batched, modals, ids, None)
    ids_full = torch.tensor([1] + [IMAGE] * 8 + [2] + [IMAGE] * 6)
    eq_mm, eq_mask = fn(ns, mm, mask, batched, modals, ids_full, torch.arange(16))
    assert same_mm is mm and eq_mm is mm
    save("truncate.npz", mm=mm, mask=mask, batched=batched, input_ids=ids, position_ids=pos, out_mm=out_mm, out_mask=out_mask,
         input_ids_full=ids_full)


def golden_kmeans():
    cases = {}
    for ci, (T, P, D, K, seed) in enumerate([(150, 3, 16, 10, 0), (64, 2, 8, 5, 1), (40, 1, 32, 6, 2), (6, 2, 4, 8, 3)]):
        g = torch.Generator().manual_seed(100 + ci)
        centers = torch.randn(K, P * D, generator=g) * 4
        lab = torch.randint(0, K, (T,), generator=g)
        feats = (centers[lab] + 0.5 * torch.randn(T, P * D, generator=g)).view(T, P, D)
        if ci == 2:
            feats = feats.bfloat16()                          # the GPU path hands bf16 features over
        ts = torch.arange(T, dtype=torch.float32) + 0.5 * torch.rand(T, generator=g)
        random.seed(seed)
        torch.manual_seed(seed)
        cf, ct, assign = ref_kmeans(feats, ts, K)
        cases[f"c{ci}_features"] = feats.float()
        cases[f"c{ci}_is_bf16"] = np.int64(ci == 2)
        cases[f"c{ci}_ts"] = ts
        cases[f"c{ci}_K"] = np.int64(K)
        cases[f"c{ci}_seed"] = np.int64(seed)
        cases[f"c{ci}_centres"] = cf.float()
        cases[f"c{ci}_centre_ts"] = ct
        if assign is not None:
            cases[f"c{ci}_assign"] = assign
            sel = ref_chat.select_additional_frames(feats, cf, assign, 2)
            cases[f"c{ci}_extra"] = torch.cat(sel).sort().values
            cases[f"c{ci}_extra_counts"] = torch.tensor([len(s) for s in sel])
    cases["n_cases"] = np.int64(4)
    save("kmeans.npz", **cases)


def golden_kmeans_reseed():
    """the empty-cluster branch of the reference (kmeans_with_time.py:116-120: one random.randint(0, T-1) per empty
    cluster, ascending cluster index, every iteration). Degenerate inputs -- a handful of distinct rows repeated, constant
    time stamps -- make duplicate centres, whose later copies stay empty: case 0 reseeds three clusters once, case 1
    seven clusters in every one of the 30 iterations (211 draws in all). Stored with the number of draws and the value
    of random.random() right after the call, so that an implementation that pre-draws reseed rows can be held to
    leaving the generator exactly where the reference leaves it."""
    cases = {}
    for ci, (T, P, D, K, seed, nd) in enumerate([(48, 2, 8, 6, 5, 3), (40, 2, 8, 10, 7, 3)]):
        g = torch.Generator().manual_seed(500 + seed)
        base = torch.randn(nd, P * D, generator=g) * 3
        lab = torch.randint(0, nd, (T,), generator=g)
        feats = base[lab].view(T, P, D).clone()
        ts = torch.zeros(T)
        calls = []
        orig = random.randint

        def counted(a, b, _orig=orig, _calls=calls):
            v = _orig(a, b)
            _calls.append(v)
            return v

        random.randint = counted
        try:
            random.seed(seed)
            torch.manual_seed(seed)
            cf, ct, assign = ref_kmeans(feats, ts, K)
            nxt = random.random()
        finally:
            random.randint = orig
        cases.update({f"c{ci}_features": feats, f"c{ci}_ts": ts, f"c{ci}_K": np.int64(K), f"c{ci}_seed": np.int64(seed),
                      f"c{ci}_centres": cf.float(), f"c{ci}_centre_ts": ct, f"c{ci}_assign": assign,
                      f"c{ci}_randint_calls": np.int64(len(calls)), f"c{ci}_next_random": np.float64(nxt)})
        print("kmeans_reseed case", ci, "randint draws", len(calls), "sizes", torch.bincount(assign, minlength=K).tolist())
    cases["n_cases"] = np.int64(2)
    save("kmeans_reseed.npz", **cases)


def golden_compress():
    g = torch.Generator().manual_seed(31)
    t, gh, gw, ms = 6, 4, 6, 2
    P, E = (gh // ms) * (gw // ms), ms * ms * 588
    base = torch.rand(1, P, E, generator=g) * 2 - 1
    pix = base.repeat(t, 1, 1)
    pix[1] += 0.002 * torch.randn(P, E, generator=g)
    pix[2, :3] += 0.5 * torch.randn(3, E, generator=g)
    pix[4] = pix[3]
    pix = pix.reshape(-1, 588)
    grid, merge = torch.tensor([[t, gh, gw]]), torch.tensor([ms])
    batched = grid.prod(dim=1).div(merge ** 2).long()
    fn = ref_chat.Videollama3MetaForCausalLM._get_compression_mask
    out = {}
    for tag, px in (("f32", pix), ("bf16", pix.bfloat16())):
        out[f"mask_{tag}"] = fn(None, px, batched, grid, merge, ["video"], minor_frame_indices=[])
        out[f"mask_minor_{tag}"] = fn(None, px, batched, grid, merge, ["video"], minor_frame_indices=[2, 5])
    mm = torch.randn(t * P, 64, generator=g)
    ev = ref_chat.Videollama3MetaForCausalLM.compress_unimportant_events(None, mm, P, [1, 4])
    ids = torch.tensor([1, 2] + sum(([70, 71] + [IMAGE] * P + [44] for _ in range(t)), []) + [9, 9, 9])
    self_ns = types.SimpleNamespace(config=types.SimpleNamespace(image_token_index=IMAGE))
    mask = out["mask_minor_f32"]
    mm2, ids2, am2, _, _ = ref_chat.Videollama3MetaForCausalLM._compress_visual_tokens(
        self_ns, mask, ev, ids, torch.ones_like(ids))
    save("compress.npz", pixel_values=pix, grid_sizes=grid, merge_sizes=merge, mm=mm, event_pooled=ev,
         input_ids=ids, ids_compressed=ids2, mm_compressed=mm2, **out)


def golden_text():
    tok = ToyTokenizer()
    img = "<image>" * 2
    sys_ = "<|im_start|>system\nYou are a helpful assistant.<|im_end|>\n"
    text = (sys_ + f"<|im_start|>user\nTime 0.0s:{img},Time 1.0s:{img}\nWhat is on the table?<|im_end|>\n"
            "<|im_start|>assistant\nA red cup.<|im_end|>\n"
            f"<|im_start|>user\nTime 2.0s:{img},Time 3.5s:{img}\nWho enters the room?<|im_end|>\n"
            "<|im_start|>assistant\nA man in a blue coat.<|im_end|>\n"
            f"<|im_start|>user\nTime 4.0s:{img}\nWhat does he pick up?<|im_end|>\n<|im_start|>assistant\n")
    hq = ["What is on the table?", "Who enters the room?"]
    ha = ["A red cup.", "A man in a blue coat."]
    cur = "What does he pick up?"
    cases = []
    for sel in ["[yes,0,1]", "[yes,1]", "[yes]", "[no,0]", "[no]", "[yes,0]", "[no,0,1]", "[yes,5]", "[maybe,1]"]:
        ns = types.SimpleNamespace(hist_qs=hq, hist_as=ha, current_question=cur, tokenizer=tok)
        new_inputs, if_visual = ref_chat.Videollama3MetaForCausalLM.prepare_inputs(ns, sel, original_text=text)
        cases.append({"selection": sel, "if_visual": bool(if_visual),
                      "prompt": tok.decode(new_inputs["input_ids"][0]), "ids": new_inputs["input_ids"][0].tolist()})
    summary = ref_chat.create_visual_summary_prompt(6, torch.tensor([0.0, 1.04, 12.5]))
    qa_prompt = ref_format_example({"current_Q": cur, "hist_Qs": hq, "hist_As": ha})
    qa_prompt_nodemo = ref_format_example({"current_Q": cur, "hist_Qs": hq, "hist_As": ha}, include_demo=False)
    with open(os.path.join(HERE, "text.json"), "w") as f:
        json.dump({"original_text": text, "hist_qs": hq, "hist_as": ha, "current_question": cur, "cases": cases,
                   "summary_prompt": summary, "qa_prompt": qa_prompt, "qa_prompt_nodemo": qa_prompt_nodemo}, f, indent=1)
    print("wrote text.json")


def build_ref_model(vst, pst, lst):
    vcfg = dict(VIT, patch_size=14)
    cfg = Videollama3Qwen2Config(vision_encoder_config=vcfg, image_token_index=IMAGE, vocab_size=VOCAB,
                                 hidden_size=LLM["hidden_size"], intermediate_size=LLM["intermediate_size"],
                                 num_hidden_layers=LLM["num_hidden_layers"], num_attention_heads=LLM["num_attention_heads"],
                                 num_key_value_heads=LLM["num_key_value_heads"], rms_norm_eps=1e-6, rope_theta=1e6,
                                 max_position_embeddings=32768, tie_word_embeddings=False, eos_token_id=IM_END,
                                 bos_token_id=259, pad_token_id=259, use_token_compression=True)
    cfg._attn_implementation = "eager"
    cfg.vision_encoder_config._attn_implementation = "eager"
    model = ref_chat.Videollama3Qwen2ForCausalLM(cfg).eval()
    sd = {}
    for k, v in vst.items():
        sd["model.vision_encoder." + k] = v
    for k, v in pst.items():
        sd["model.mm_projector." + k] = v
    for k, v in lst.items():
        sd[("lm_head.weight" if k == "lm_head.weight" else "model." + k)] = v
    res = model.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys, res.unexpected_keys
    assert all("rotary" in k or "inv_freq" in k for k in res.missing_keys), res.missing_keys
    return model


def golden_e2e():
    """full reference pipeline, tiny model, toy tokenizer, greedy: (a) 8 frames + 2 history turns (retrieval +
    prompt surgery), (b) 150 frames (k-means, K=10 event passes, event compression). The reference's CPU path is
    the eager/global-attention encoder."""
    vst = random_vit_state(VisionConfig(**VIT), seed=3, std=0.05)
    pst = random_proj_state(VIT["hidden_size"], LLM["hidden_size"], seed=1, std=0.05)
    lst = random_llm_state(LlmConfig(**LLM), seed=7, std=0.05)
    model = build_ref_model(vst, pst, lst)
    tok = ToyTokenizer()
    out = {"vit_checksum": np.float64(checksum(vst)), "llm_checksum": np.float64(checksum(lst))}
    for tag, T, hist in (("a", 8, True), ("b", 150, False), ("c", 150, False)):
        inp = e2e_inputs(tag)
        pix, grid, merge, ts, text = inp["pixel_values"], inp["grid_sizes"], inp["merge_sizes"], inp["timestamps"], inp["text"]
        hq, ha, cur = inp["hist_qs"], inp["hist_as"], inp["current_question"]
        enc = tok(text)
        random.seed(5)
        torch.manual_seed(5)
        with torch.no_grad():
            sel = model.qa_selection(current_question=cur, hist_qs=hq, hist_as=ha, tokenizer=tok, original_text=text,
                                     input_ids=enc["input_ids"], attention_mask=enc["attention_mask"], mode="FCC",
                                     all_timestamps=ts)
            # instrument: capture the intermediate products of prepare_inputs_labels_for_multimodal
            cap = {}
            orig_sel = model.select_events_based_on_summary
            orig_mask = model._get_compression_mask

            def wrap_sel(mm, n, t_):
                r = orig_sel(mm, n, t_)
                cap["minor"] = list(r)
                cap["mm_features"] = mm.clone()
                return r

            def wrap_mask(*a, **k):
                r = orig_mask(*a, **k)
                cap["mask"] = r.clone()
                return r

            model.select_events_based_on_summary = wrap_sel
            model._get_compression_mask = wrap_mask
            # capture (b) / force (c) the event-vs-question cosines and capture the cluster assignments
            orig_cos, orig_km = ref_chat.F.cosine_similarity, ref_chat.kmeans_with_time_min_max
            forced = torch.tensor(FORCED_COSINE)

            def wrap_cos(a, b, dim=1):
                r = orig_cos(a, b, dim=dim)
                cap["cosine"] = r.clone()
                return forced.to(r.dtype) if tag == "c" else r

            def wrap_km(f, t_, k):
                r = orig_km(f, t_, k)
                cap["assign"] = r[2].clone()
                return r

            ref_chat.F.cosine_similarity = wrap_cos
            ref_chat.kmeans_with_time_min_max = wrap_km
            new_ids, sel_str = model.generate(pixel_values=pix, grid_sizes=grid, merge_sizes=merge, modals=["video"],
                                              new_input_ids=sel["new_input_ids"], new_attention_mask=sel["new_attention_mask"],
                                              selection_module_output=sel["selection_module_output"],
                                              if_visual=sel["if_visual"], total_image_num=T, max_new_tokens=8,
                                              do_sample=False, repetition_penalty=1.05)
            model.select_events_based_on_summary = orig_sel
            model._get_compression_mask = orig_mask
            ref_chat.F.cosine_similarity, ref_chat.kmeans_with_time_min_max = orig_cos, orig_km
        out.update({f"{tag}_pix_checksum": np.float64(pix.double().abs().sum()), f"{tag}_T": np.int64(T),
                    f"{tag}_selection": np.array(sel["selection_module_output"]),
                    f"{tag}_if_visual": np.int64(bool(sel["if_visual"])), f"{tag}_new_input_ids": sel["new_input_ids"][0],
                    f"{tag}_minor": np.array(cap.get("minor", []), dtype=np.int64),
                    f"{tag}_mask": cap["mask"] if "mask" in cap else np.zeros(0, dtype=bool),
                    f"{tag}_mm_features": cap.get("mm_features", torch.zeros(0)),
                    f"{tag}_tokens": new_ids[0], f"{tag}_sel_out": np.array(sel_str),
                    f"{tag}_cosine": cap.get("cosine", torch.zeros(0)), f"{tag}_assign": cap.get("assign", torch.zeros(0, dtype=torch.long)),
                    })
        print(tag, "selection:", repr(sel["selection_module_output"]), "minor frames:", len(cap.get("minor", [])),
              "kept tokens:", int(cap["mask"].sum()) if "mask" in cap else None, "new tokens:", new_ids[0].tolist())
    save("e2e.npz", **out)


def golden_e2e_blockdiag():
    """cases b and c of golden_e2e with the encoder in the semantics the reference SHIPS with on a GPU
    (flash_attention_2: per-frame block-diagonal attention, modeling_videollama3_encoder.py:309-312). flash_attn is
    not installable here, so the reference's eager encoder is called ONE FRAME PER CALL inside the reference's own
    encode_images (the +1 same-frame bias is constant over a single frame, hence softmax-invariant: golden_vit's
    construction). This is the mode the frame-sharded encoder runs in, so the N-rank pipeline
    tests are checked against it. Everything downstream -- k-means, event passes, compression, generate -- is the
    reference unchanged."""
    vst = random_vit_state(VisionConfig(**VIT), seed=3, std=0.05)
    pst = random_proj_state(VIT["hidden_size"], LLM["hidden_size"], seed=1, std=0.05)
    lst = random_llm_state(LlmConfig(**LLM), seed=7, std=0.05)
    model = build_ref_model(vst, pst, lst)
    venc = model.get_model().get_vision_encoder()
    whole = venc.forward

    def per_frame(pixel_values, grid_sizes, merge_sizes):
        outs, row = [], 0
        for (t, h, w), ms in zip(grid_sizes.tolist(), merge_sizes.tolist()):
            for _ in range(t):
                outs.append(whole(pixel_values[row:row + h * w], torch.tensor([[1, h, w]]), torch.tensor([ms])))
                row += h * w
        return torch.cat(outs, 0)

    venc.forward = per_frame
    tok = ToyTokenizer()
    out = {"vit_checksum": np.float64(checksum(vst)), "llm_checksum": np.float64(checksum(lst))}
    for tag in ("b", "c"):
        inp = e2e_inputs(tag)
        enc = tok(inp["text"])
        random.seed(5)
        torch.manual_seed(5)
        cap = {}
        sel = model.qa_selection(current_question=inp["current_question"], hist_qs=inp["hist_qs"], hist_as=inp["hist_as"],
                                 tokenizer=tok, original_text=inp["text"], input_ids=enc["input_ids"],
                                 attention_mask=enc["attention_mask"], mode="FCC", all_timestamps=inp["timestamps"])
        orig_sel, orig_mask = model.select_events_based_on_summary, model._get_compression_mask
        orig_cos, orig_km = ref_chat.F.cosine_similarity, ref_chat.kmeans_with_time_min_max
        forced = torch.tensor(FORCED_COSINE)

        def wrap_sel(mm, n, t_):
            r = orig_sel(mm, n, t_)
            cap["minor"] = list(r)
            return r

        def wrap_mask(*a, **k):
            r = orig_mask(*a, **k)
            cap["mask"] = r.clone()
            return r

        def wrap_cos(a, b, dim=1):
            r = orig_cos(a, b, dim=dim)
            cap["cosine"] = r.clone()
            return forced.to(r.dtype) if tag == "c" else r

        def wrap_km(f, t_, k):
            r = orig_km(f, t_, k)
            cap["assign"] = r[2].clone()
            return r

        model.select_events_based_on_summary, model._get_compression_mask = wrap_sel, wrap_mask
        ref_chat.F.cosine_similarity, ref_chat.kmeans_with_time_min_max = wrap_cos, wrap_km
        try:
            new_ids, sel_str = model.generate(pixel_values=inp["pixel_values"], grid_sizes=inp["grid_sizes"],
                                              merge_sizes=inp["merge_sizes"], modals=["video"],
                                              new_input_ids=sel["new_input_ids"], new_attention_mask=sel["new_attention_mask"],
                                              selection_module_output=sel["selection_module_output"],
                                              if_visual=sel["if_visual"], total_image_num=inp["T"], max_new_tokens=8,
                                              do_sample=False, repetition_penalty=1.05)
        finally:
            model.select_events_based_on_summary, model._get_compression_mask = orig_sel, orig_mask
            ref_chat.F.cosine_similarity, ref_chat.kmeans_with_time_min_max = orig_cos, orig_km
        out.update({f"{tag}_pix_checksum": np.float64(inp["pixel_values"].double().abs().sum()),
                    f"{tag}_minor": np.array(cap["minor"], dtype=np.int64), f"{tag}_mask": cap["mask"],
                    f"{tag}_tokens": new_ids[0], f"{tag}_cosine": cap["cosine"], f"{tag}_assign": cap["assign"]})
        print(tag, "block-diagonal: minor frames", len(cap["minor"]), "kept tokens", int(cap["mask"].sum()),
              "new tokens", new_ids[0].tolist())
    save("e2e_blockdiag.npz", **out)


QWEN2_CASES = {
    # tag: (LlmConfig overrides, prompt length)  -- "t" = the tiny e2e model; "g" = GQA with 3 query heads per kv head
    "t": (dict(LLM), 37),
    "g": (dict(LLM, hidden_size=768, intermediate_size=640, num_attention_heads=6, num_key_value_heads=2,
               num_hidden_layers=3), 150),
}


def golden_qwen2():
    """the Qwen2 seams the reference calls, run on the REFERENCE's own model object
    (Videollama3Qwen2ForCausalLM, cogreasoner_chat.py:591-598; transformers' Qwen2 arithmetic underneath):
    get_model()(inputs_embeds=, attention_mask=).last_hidden_state (:312-316,322), its mean over the sequence
    (:317,323), lm_head logits of the last row and of three cached single-token steps (the generate() loop,
    :802-807). Stored for the fp32 model and for the same model cast to bf16 (how the reference runs on a GPU,
    evaluate/answer_generate.py:176), both on CPU. The three decode steps are fed the fp32 model's greedy tokens
    in both precisions so that every stored vector has the same inputs."""
    out = {"transformers_version": np.array(__import__("transformers").__version__)}
    for tag, (llm, S) in QWEN2_CASES.items():
        lcfg = LlmConfig(**llm)
        lst = random_llm_state(lcfg, seed=7, std=0.05)
        vst = random_vit_state(VisionConfig(**VIT), seed=3, std=0.05)
        pst = random_proj_state(VIT["hidden_size"], llm["hidden_size"], seed=1, std=0.05)
        saved = dict(LLM)
        LLM.clear(); LLM.update(llm)
        try:
            model = build_ref_model(vst, pst, lst)
        finally:
            LLM.clear(); LLM.update(saved)
        g = torch.Generator().manual_seed(900 + S)
        embeds = torch.randn(S, llm["hidden_size"], generator=g) * 0.5
        out[f"{tag}_embeds"] = embeds
        out[f"{tag}_llm_checksum"] = np.float64(checksum(lst))
        out[f"{tag}_cfg"] = np.array(json.dumps(llm))
        steps = None
        for prec, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
            m = copy.deepcopy(model).to(dt).eval()
            e = embeds.to(dt)[None]
            with torch.no_grad():
                r = m.get_model()(inputs_embeds=e, attention_mask=torch.ones(1, S, dtype=torch.long), use_cache=True)
                hid = r.last_hidden_state[0]
                lg = [m.lm_head(hid[-1:]).float()[0]]
                pooled = torch.mean(r.last_hidden_state, dim=1)[0]
                past = r.past_key_values
                toks = []
                for i in range(3):
                    tok = int(lg[-1].argmax()) if steps is None else steps[i]
                    toks.append(tok)
                    te = m.get_model().embed_tokens(torch.tensor([[tok]]))
                    r = m.get_model()(inputs_embeds=te, attention_mask=torch.ones(1, S + i + 1, dtype=torch.long),
                                      past_key_values=past, use_cache=True)
                    past = r.past_key_values
                    lg.append(m.lm_head(r.last_hidden_state[0, -1:]).float()[0])
            if steps is None:
                steps = toks
                out[f"{tag}_step_tokens"] = np.array(toks, dtype=np.int64)
            out[f"{tag}_hidden_{prec}"] = hid.float()
            out[f"{tag}_pooled_{prec}"] = pooled.float()
            out[f"{tag}_logits_{prec}"] = torch.stack(lg)          # [4, vocab]: prefill, then 3 cached steps
        print(tag, "steps", steps, "bf16-vs-f32 logits rel",
              float((out[f"{tag}_logits_bf16"] - out[f"{tag}_logits_f32"]).abs().max() / out[f"{tag}_logits_f32"].abs().max()))
    save("qwen2.npz", **out)


class _PeftStyleLoRALinear(torch.nn.Module):
    """peft 0.15.2 `lora.Linear.forward` in eval mode, written out with nn.Linear modules (peft itself is not installed in
    this image and cannot be): result = base(x) + lora_B(lora_A(dropout(x))) * (lora_alpha / r), dropout = identity in
    eval mode, both low-rank maps bias-free (LoraConfig(bias="none"), train/second_stage_training.py:257-264)."""

    def __init__(self, base: torch.nn.Linear, A: torch.Tensor, B: torch.Tensor, alpha: float):
        super().__init__()
        r = A.shape[0]
        self.base_layer = base
        self.lora_dropout = torch.nn.Dropout(p=0.1)
        self.lora_A = torch.nn.Linear(base.in_features, r, bias=False)
        self.lora_B = torch.nn.Linear(r, base.out_features, bias=False)
        self.lora_A.weight.data.copy_(A)
        self.lora_B.weight.data.copy_(B)
        self.scaling = alpha / r

    def forward(self, x):
        return self.base_layer(x) + self.lora_B(self.lora_A(self.lora_dropout(x))) * self.scaling


def golden_lora():
    """LoRA branch. The REFERENCE's model object (Videollama3Qwen2ForCausalLM, tiny) with the
    linears of the reference's target-module list (train/second_stage_training.py:241-254: q/k/v/o/gate/up/down of
    every decoder layer + mm_projector.readout.0/.2) wrapped by _PeftStyleLoRALinear, r = 8, lora_alpha = 16 (:257-264),
    eval mode. Stored: the projector output and the Qwen2 last_hidden_state / last-row logits for the base model and for
    two adapters (the reference loads two: 'full_module' and 'language_module', evaluate/answer_generate.py:181-182).
    peft itself could not be run here (not installed, no network): this pins the restated branch of oracle/qwen2.py and
    the merged-weight path of cogstream_amd.weights.merge_lora to an evaluation by independent modules inside the
    reference's own model, not to peft's code."""
    from cogstream_amd.weights import random_lora_state
    lcfg, vcfg = LlmConfig(**LLM), VisionConfig(**VIT)
    lst = random_llm_state(lcfg, seed=7, std=0.05)
    vst = random_vit_state(vcfg, seed=3, std=0.05)
    pst = random_proj_state(VIT["hidden_size"], LLM["hidden_size"], seed=1, std=0.05)
    g = torch.Generator().manual_seed(4242)
    S, M = 37, 24
    embeds = torch.randn(S, LLM["hidden_size"], generator=g) * 0.5
    vis = torch.randn(M, VIT["hidden_size"], generator=g) * 0.5
    targets = []
    for i in range(LLM["num_hidden_layers"]):                     # second_stage_training.py:241-250
        targets.extend([f"model.layers.{i}.self_attn.q_proj", f"model.layers.{i}.self_attn.k_proj",
                        f"model.layers.{i}.self_attn.v_proj", f"model.layers.{i}.self_attn.o_proj",
                        f"model.layers.{i}.mlp.gate_proj", f"model.layers.{i}.mlp.up_proj", f"model.layers.{i}.mlp.down_proj"])
    targets.extend(["model.mm_projector.readout.0", "model.mm_projector.readout.2"])       # :251-254
    out = {"embeds": embeds, "vis": vis, "targets": np.array(json.dumps(targets)), "r": np.int64(8), "lora_alpha": np.float64(16.0),
           "llm_checksum": np.float64(checksum(lst))}

    def run(model, tag):
        with torch.no_grad():
            r = model.get_model()(inputs_embeds=embeds[None], attention_mask=torch.ones(1, S, dtype=torch.long))
            hid = r.last_hidden_state[0]
            out[f"{tag}_hidden"] = hid
            out[f"{tag}_logits"] = model.lm_head(hid[-1:])[0]
            out[f"{tag}_projected"] = model.get_model().mm_projector(vis)

    run(build_ref_model(vst, pst, lst), "base")
    for a, seed in (("full_module", 21), ("language_module", 22)):
        lora = random_lora_state(lcfg, seed=seed, r=8, proj_dims=(VIT["hidden_size"], LLM["hidden_size"]))
        out[f"{a}_seed"] = np.int64(seed)
        out[f"{a}_checksum"] = np.float64(checksum(lora))
        model = build_ref_model(vst, pst, lst)
        n_wrapped = 0
        for name in targets:
            parent_name, _, leaf = name.rpartition(".")
            parent = model.get_submodule(parent_name)
            base = getattr(parent, leaf) if not leaf.isdigit() else parent[int(leaf)]
            assert isinstance(base, torch.nn.Linear), name
            key = "base_model.model." + name
            wrapped = _PeftStyleLoRALinear(base, lora[key + ".lora_A.weight"], lora[key + ".lora_B.weight"], 16.0)
            if leaf.isdigit():
                parent[int(leaf)] = wrapped
            else:
                setattr(parent, leaf, wrapped)
            n_wrapped += 1
        assert n_wrapped == len(targets) == 7 * LLM["num_hidden_layers"] + 2 and 2 * n_wrapped == len(lora)
        run(model.eval(), a)
        print(a, "logits moved by", float((out[f"{a}_logits"] - out["base_logits"]).abs().max() / out["base_logits"].abs().max()))
    save("lora.npz", **out)


def golden_vit_bf16():
    """the encoder + projector cast to bf16 (evaluate/answer_generate.py:176), block-diagonal semantics (one frame
    per call of the eager path = the flash path), on the inputs of vit_tiny.npz rounded to bf16 (:70)."""
    cfg = VisionConfig(**VIT)
    st = random_vit_state(cfg, seed=3, std=0.05)
    m = ref_vit(st, cfg).to(torch.bfloat16)
    gold = np.load(os.path.join(HERE, "vit_tiny.npz"))
    pix = torch.from_numpy(gold["pixel_values"]).bfloat16()
    grid, merge = torch.from_numpy(gold["grid_sizes"]), torch.from_numpy(gold["merge_sizes"])
    outs, row = [], 0
    with torch.no_grad():
        for (t, h, w), ms in zip(grid.tolist(), merge.tolist()):
            for f in range(t):
                outs.append(m(pix[row:row + h * w], torch.tensor([[1, h, w]]), torch.tensor([ms])))
                row += h * w
        block_diag = torch.cat(outs, 0)
        pst = random_proj_state(cfg.hidden_size, LLM["hidden_size"], seed=1, std=0.05)
        cfg_ns = types.SimpleNamespace(vision_encoder_config=types.SimpleNamespace(hidden_size=cfg.hidden_size),
                                       hidden_size=LLM["hidden_size"])
        proj = ref_chat.MlpGeluProjector(cfg_ns, "mlp2x_gelu").eval()
        proj.load_state_dict(pst, strict=True)
        projected = proj.to(torch.bfloat16)(block_diag)
    f32 = torch.from_numpy(gold["block_diag"])
    print("vit bf16 vs f32 reference: rel max", float((block_diag.float() - f32).abs().max() / f32.abs().max()))
    save("vit_tiny_bf16.npz", block_diag=block_diag.float(), projected=projected.float())


def _rle(ids):
    """[[id, run], ...] -- prompts are dominated by runs of the <image> id"""
    out = []
    for i in ids:
        if out and out[-1][0] == i:
            out[-1][1] += 1
        else:
            out.append([int(i), 1])
    return out


def golden_tokenizer():
    """Facts about the REAL Qwen2 tokenizer of the checkpoint (/root/reference/model/{vocab.json,merges.txt,
    added_tokens.json,tokenizer_config.json}: data) and the reference code driven with it:
      * the allowed-id set of StructuredLogitsProcessor (qaselect_module_predict.py:86-103), read off the processor
        object the reference's select_qas hands to generate_language_module;
      * model/chat_template.json rendered by the tokenizer's jinja engine for a set of conversations
        (processing_cogreasoner.py:752-801 forwards to tokenizer.apply_chat_template);
      * token ids of the SURVEY appendix-B3 strings and of the cfg2 (64 frames x 231 tokens) / cfg4 prompts;
      * prepare_inputs / process_input_ids (cogreasoner_chat.py:121-177,478-511) on a real-tokenizer prompt.
    Every (text, ids) pair that was tokenised is recorded so that tests can replay the tokenizer by lookup."""
    from transformers import Qwen2Tokenizer
    from model.qaselect_module_predict import select_qas as ref_select_qas
    tok = Qwen2Tokenizer.from_pretrained("/root/reference/model")
    ct = json.load(open("/root/reference/model/chat_template.json"))["chat_template"]
    pairs = []

    class Recording:
        """the real tokenizer, recording every encode / decode that goes through it"""
        def __call__(self, text, **kw):
            r = tok(text, **kw)
            ids = r["input_ids"]
            pairs.append((text, (ids[0] if hasattr(ids, "shape") and ids.ndim == 2 else ids)))
            return r

        def encode(self, text, **kw):
            ids = tok.encode(text, **kw)
            pairs.append((text, ids))
            return ids

        def decode(self, ids, **kw):
            text = tok.decode(ids, **kw)
            if not kw.get("skip_special_tokens"):
                pairs.append((text, ids))
            return text

        def __getattr__(self, k):
            return getattr(tok, k)

    rec = Recording()
    out = {"vocab_size": len(tok), "image_token_id": tok.convert_tokens_to_ids("<image>"),
           "im_start": tok.convert_tokens_to_ids("<|im_start|>"), "im_end": tok.convert_tokens_to_ids("<|im_end|>"),
           "endoftext": tok.convert_tokens_to_ids("<|endoftext|>")}

    # --- allowed ids, from the reference's own processor object
    cap = {}

    class FakeModel:
        device = torch.device("cpu")

        def generate_language_module(self, **kw):
            cap["allowed"] = list(kw["logits_processor"][0].allowed_token_ids)
            cap["kw"] = {k: v for k, v in kw.items() if k in ("max_new_tokens", "num_beams", "do_sample", "eos_token_id")}
            gen = torch.tensor(tok.encode("[yes,0]") + [151645])
            return torch.cat([kw["input_ids"][0], gen])[None]

    hq = ["What is on the table?", "Who enters the room?"]
    ha = ["A red cup.", "A man in a blue coat."]
    cur = "What does he pick up?"
    sel = ref_select_qas(cur, hq, ha, FakeModel(), tokenizer=rec)
    out["allowed_ids"] = cap["allowed"]
    out["select_kwargs"] = cap["kw"]
    out["select_roundtrip"] = sel
    out["qa_prompt_len"] = len(tok(ref_format_example({"current_Q": cur, "hist_Qs": hq, "hist_As": ha}))["input_ids"])

    # --- chat template renderings (jinja on the reference's template)
    convs = {
        "b3": [{"role": "system", "content": "You are a helpful assistant."},
               {"role": "user", "content": [{"type": "video", "num_frames": 3, "timestamps": [0.0, 1.0, 2.04]},
                                            {"type": "text", "text": "Q1?"}]},
               {"role": "assistant", "content": "A1."}, {"role": "user", "content": "Q2?"}],
        "default_system": [{"role": "user", "content": [{"type": "video", "num_frames": 2, "timestamps": [12.0, 13.06]},
                                                        {"type": "text", "text": "What is happening in the video?"}]}],
        "image_and_stream": [{"role": "user", "content": [{"type": "image", "timestamp": 3.25}, {"type": "image"},
                                                          {"type": "video", "num_frames": 2}, "plain string item",
                                                          {"type": "text", "text": "Describe."}]},
                             {"role": "stream", "content": [{"type": "text", "text": "partial"}]},
                             {"role": "assistant", "content": "ok"}],
    }
    out["templates"] = []
    for name, conv in convs.items():
        for sysp in (True, False):
            for gen in (True, False):
                txt = tok.apply_chat_template(conv, chat_template=ct, tokenize=False, add_system_prompt=sysp,
                                              add_generation_prompt=gen, image_token="<image>")
                out["templates"].append({"name": name, "conversation": conv, "add_system_prompt": sysp,
                                         "add_generation_prompt": gen, "text": txt})

    # --- appendix B3 strings
    out["strings"] = {s: rec.encode(s, add_special_tokens=False)
                      for s in ["Time 12.0s:", ",", "\n", "<|im_start|>assistant\n", "<image>", "[yes,0,5]", "[no]",
                                "yes", "no", "[", "]", "<|im_end|>"] + [str(i) for i in range(10)]}

    # --- prompt lengths: cfg2 (64 frames x 231) and the cfg4 session (8 turns, 8 frames per turn, P = 256 .. 231)
    def video_prompt(T, P, question, history=()):
        conv = []
        for q, a in history:
            conv += [{"role": "user", "content": q}, {"role": "assistant", "content": a}]
        conv.append({"role": "user", "content": [{"type": "video", "num_frames": T, "timestamps": [float(i) for i in range(T)]},
                                                 {"type": "text", "text": question}]})
        txt = tok.apply_chat_template(conv, chat_template=ct, tokenize=False, add_system_prompt=True,
                                      add_generation_prompt=True, image_token="<image>")
        txt = txt.replace("<image>", "<image>" * P)
        return txt, rec(txt)["input_ids"]

    q = "What is happening in the video?"
    out["prompts"] = []
    for T, P in ((64, 231), (8, 64), (256, 50)):
        txt, ids = video_prompt(T, P, q)
        out["prompts"].append({"T": T, "P": P, "question": q, "len": len(ids), "ids_rle": _rle(ids),
                               "n_image": int(sum(1 for i in ids if i == out["image_token_id"])),
                               "system_len": len(tok.encode(txt.split("<|im_start|>user")[0]))})
    hist = json.load(open(os.path.join(HERE, "cfg4_history.json")))
    out["cfg4_retrieval_prompt_lens"] = []
    hqs, has = [t["question"] for t in hist["turns"]], [t["answer"] for t in hist["turns"]]
    for n in range(1, len(hqs) + 1):
        p = ref_format_example({"current_Q": hqs[n - 1], "hist_Qs": hqs[:n - 1], "hist_As": has[:n - 1]})
        out["cfg4_retrieval_prompt_lens"].append(len(rec(p)["input_ids"]))

    # --- prompt surgery with the real tokenizer (the reference's prepare_inputs decodes, edits and re-tokenises)
    img = "<image>" * 4
    text = ("<|im_start|>system\nYou are a helpful assistant.<|im_end|>\n"
            f"<|im_start|>user\nTime 0.0s:{img},Time 1.0s:{img}\n{hq[0]}<|im_end|>\n<|im_start|>assistant\n{ha[0]}<|im_end|>\n"
            f"<|im_start|>user\nTime 2.0s:{img},Time 3.5s:{img}\n{hq[1]}<|im_end|>\n<|im_start|>assistant\n{ha[1]}<|im_end|>\n"
            f"<|im_start|>user\nTime 14.0s:{img}\n{cur}<|im_end|>\n<|im_start|>assistant\n")
    out["surgery"] = {"original_text": text, "hist_qs": hq, "hist_as": ha, "current_question": cur, "cases": []}
    for s in ["[yes,0,1]", "[yes,1]", "[yes]", "[no,0]", "[no]", "[no,0,1]", "[yes,5]"]:
        ns = types.SimpleNamespace(hist_qs=hq, hist_as=ha, current_question=cur, tokenizer=rec)
        new_inputs, if_visual = ref_chat.Videollama3MetaForCausalLM.prepare_inputs(ns, s, original_text=text)
        ids = new_inputs["input_ids"][0].tolist()
        out["surgery"]["cases"].append({"selection": s, "if_visual": bool(if_visual), "ids_rle": _rle(ids),
                                        "prompt": tok.decode(ids)})
    # event-summary prompt (cogreasoner_chat.py:93-119,297-298) through the real tokenizer
    sp = ref_chat.create_visual_summary_prompt(15 * 50, torch.arange(15, dtype=torch.float32) + 30)
    out["summary_prompt_len"] = len(rec(sp)["input_ids"])
    out["pairs"] = [{"text": t, "ids_rle": _rle(list(map(int, (i.tolist() if hasattr(i, "tolist") else i))))} for t, i in pairs]
    with open(os.path.join(HERE, "tokenizer.json"), "w") as f:
        json.dump(out, f, ensure_ascii=False, indent=None, separators=(",", ":"))
    print("tokenizer.json: allowed", out["allowed_ids"], "cfg2 prompt", out["prompts"][0]["len"], "pairs", len(out["pairs"]),
          os.path.getsize(os.path.join(HERE, "tokenizer.json")), "bytes")


def golden_sampling():
    """HF logits processors / warpers in the order GenerationMixin applies them for the reference's shipped
    generation_config.json (do_sample, temperature 0.7, top_k 20, top_p 0.8, repetition_penalty 1.05;
    evaluate/answer_generate.py:74 does not override it), the CPU multinomial draw, and a SAMPLED run of the
    reference's whole pipeline on the tiny model. transformers here is 5.15 (the reference pins 4.46.3; the four
    processor classes are unchanged in behaviour)."""
    from transformers.generation.logits_process import (RepetitionPenaltyLogitsProcessor, TemperatureLogitsWarper,
                                                        TopKLogitsWarper, TopPLogitsWarper)
    gc = json.load(open("/root/reference/model/generation_config.json"))
    V = 4096
    out = {"generation_config": np.array(json.dumps(gc)), "n_cases": np.int64(0)}
    g = torch.Generator().manual_seed(77)
    cases = []
    for ci, (scale, top_k, top_p, temp, rep) in enumerate([(3.0, 20, 0.8, 0.7, 1.05), (0.5, 20, 0.8, 0.7, 1.05),
                                                          (6.0, 20, 0.8, 0.7, 1.05), (2.0, 0, 0.8, 0.7, 1.05),
                                                          (2.0, 50, 1.0, 1.0, 1.0), (2.0, 5, 0.3, 1.3, 1.2),
                                                          (1.0, 0, 0.95, 1.0, 1.0)]):
        lg = torch.randn(1, V, generator=g) * scale
        if ci == 2:
            lg[0, 100:104] = lg[0].max() + 1.0          # exact ties at the top
        prev = torch.randint(0, V, (1, 37), generator=g)
        prev[0, :5] = lg[0].topk(5).indices              # penalise the leaders
        s = lg.clone()
        if rep != 1.0:
            s = RepetitionPenaltyLogitsProcessor(rep)(prev, s)
        out[f"s{ci}_after_penalty"] = s[0].clone()
        if temp != 1.0:
            s = TemperatureLogitsWarper(temp)(prev, s)
        if top_k:
            s = TopKLogitsWarper(top_k)(prev, s)
        kept_k = torch.isfinite(s[0]).nonzero().flatten()
        if top_p < 1.0:
            s = TopPLogitsWarper(top_p)(prev, s)
        kept = torch.isfinite(s[0]).nonzero().flatten()
        probs = torch.softmax(s, dim=-1)[0]
        draws = []
        for seed in range(16):
            gg = torch.Generator().manual_seed(1000 + seed)
            draws.append(int(torch.multinomial(probs, 1, generator=gg)))
        out.update({f"s{ci}_logits": lg[0], f"s{ci}_prev": prev[0], f"s{ci}_params": np.array([top_k, top_p, temp, rep], dtype=np.float64),
                    f"s{ci}_kept_after_topk": kept_k, f"s{ci}_kept": kept, f"s{ci}_probs": probs[kept], f"s{ci}_draws": np.array(draws)})
        cases.append((ci, len(kept_k), len(kept)))
    out["n_cases"] = np.int64(len(cases))
    print("sampling cases (case, kept after top-k, kept after top-p):", cases)

    # sampled generation of the reference pipeline (tiny model, case "a" of the e2e inputs), global CPU generator
    vst = random_vit_state(VisionConfig(**VIT), seed=3, std=0.05)
    pst = random_proj_state(VIT["hidden_size"], LLM["hidden_size"], seed=1, std=0.05)
    lst = random_llm_state(LlmConfig(**LLM), seed=7, std=0.05)
    model = build_ref_model(vst, pst, lst)
    tok = ToyTokenizer()
    inp = e2e_inputs("a")
    enc = tok(inp["text"])
    with torch.no_grad():
        sel = model.qa_selection(current_question=inp["current_question"], hist_qs=[], hist_as=[], tokenizer=tok,
                                 original_text=inp["text"], input_ids=enc["input_ids"], attention_mask=enc["attention_mask"],
                                 mode="NC", all_timestamps=inp["timestamps"])
        for seed in (11, 12):
            random.seed(seed)
            torch.manual_seed(seed)
            ids, _ = model.generate(pixel_values=inp["pixel_values"], grid_sizes=inp["grid_sizes"], merge_sizes=inp["merge_sizes"],
                                    modals=["video"], new_input_ids=sel["new_input_ids"], new_attention_mask=sel["new_attention_mask"],
                                    selection_module_output=sel["selection_module_output"], if_visual=sel["if_visual"],
                                    total_image_num=inp["T"], max_new_tokens=12, do_sample=True, temperature=gc["temperature"] * 4,
                                    top_k=gc["top_k"], top_p=gc["top_p"], repetition_penalty=gc["repetition_penalty"],
                                    eos_token_id=[IM_END])
            out[f"gen_seed{seed}"] = ids[0]
            print("sampled tokens, seed", seed, ids[0].tolist())
    out["gen_temperature"] = np.float64(gc["temperature"] * 4)
    save("sampling.npz", **out)


def golden_checkpoint_index():
    """the checkpoint directory's metadata (data, no weights exist in the repository): tensor names -> shard of
    model/model.safetensors.index.json, its total_size, and the four JSON configs from_pretrained reads"""
    m = "/root/reference/model/"
    idx = json.load(open(m + "model.safetensors.index.json"))
    out = {"total_size": idx["metadata"]["total_size"], "weight_map": idx["weight_map"]}
    for n in ("config.json", "generation_config.json", "preprocessor_config.json", "processor_config.json"):
        out[n] = json.load(open(m + n))
    with open(os.path.join(HERE, "checkpoint_index.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("checkpoint_index.json:", len(out["weight_map"]), "tensors,", out["total_size"], "bytes")


def golden_video_io():
    """Videollama3Qwen2Processor.load_video / _load_multimodal_data (model/processing_cogreasoner.py:326-509) with the
    decoder replaced: this image has no ffmpeg / cv2 / imageio / decord, so those imports are empty modules and the
    `ffmpeg` calls the two methods make are answered by cogstream_amd.video_io.select_frames (WHICH frames come out
    of the filter graph is therefore NOT pinned -- the arithmetic around it is: durations, timestamp grids,
    max_frames subsampling, temporal padding, per-content windows, the running offset between segments)."""
    import types
    import transformers.image_utils as _iu
    import transformers.video_utils as _vu
    if not hasattr(_iu, "VideoInput"):
        _iu.VideoInput = _vu.VideoInput
    for name in ("cv2", "ffmpeg", "imageio"):
        sys.modules.setdefault(name, types.ModuleType(name))
    if "decord" not in sys.modules:
        d = types.ModuleType("decord")
        d.VideoReader, d.cpu = object, (lambda *a, **k: None)
        sys.modules["decord"] = d
    import model.processing_cogreasoner as P
    from cogstream_amd import video_io as vio

    specs = {   # path -> (n native frames, native fps, stream start, container duration)
        "a.mp4": (75, 25.0, 0.0, 3.0), "b.mp4": (131, 30.0, 0.021, 4.4), "c.mp4": (250, 25.0, 0.0, 10.0),
        "d.mp4": (40, 10.0, 0.5, 4.0)}
    vids = {}
    for k, (n, f, st, du) in specs.items():
        fr = np.zeros((n, 4, 6, 3), np.uint8)
        fr[:, 0, 0, 0] = np.arange(n) % 256        # frame number in one pixel: identifies the chosen frames
        vids[k] = vio.DecodedVideo(fr, f, st, du)

    ff = P.ffmpeg
    ff.probe = lambda path: vids[path].probe()
    ff.input = lambda path, **kw: {"path": path, "input": kw, "trim": 0.0, "fps": None}
    ff.trim = lambda st, start, end: {**st, "trim": start}
    ff.setpts = lambda st, expr: st
    ff.filter = lambda st, name, *a, **kw: ({**st, "fps": kw["fps"]} if name == "fps" else st)
    ff.output = lambda st, *a, **kw: st

    def run(st, capture_stdout=True, quiet=True):
        v = vids[st["path"]]
        do_trim = "t" in st["input"]
        dur = st["input"].get("t", float(v.probe()["format"]["duration"]))
        idx = vio.select_frames(v, dur, do_trim, st["trim"], st["fps"])
        return v.frames[idx].tobytes(), None
    ff.run = run

    me = types.SimpleNamespace(fps=1, max_frames=180)
    me.load_video = lambda **kw: P.Videollama3Qwen2Processor.load_video(me, **kw)
    me.load_images = None
    cases = []
    for kw in (dict(video_path="a.mp4", fps=1, max_frames=180), dict(video_path="b.mp4", fps=2, max_frames=180),
               dict(video_path="c.mp4", fps=1, max_frames=4), dict(video_path="c.mp4", fps=1, max_frames=180, start_time=2.0, end_time=7.5),
               dict(video_path="d.mp4", fps=1, max_frames=180, start_time=0.2), dict(video_path="b.mp4", fps=1, max_frames=180, trim_time=0.6),
               dict(video_path="c.mp4", fps=2, max_frames=180, temporal_factor=4), dict(video_path="a.mp4", fps=None, max_frames=10)):
        frames, ts, dur = me.load_video(**kw)
        cases.append({"args": kw, "frame_ids": [int(f[0, 0, 0]) for f in frames], "timestamps": [float(t) for t in ts],
                      "duration": float(dur)})
    conv = [{"role": "system", "content": "You are a helpful assistant."},
            {"role": "user", "content": [{"type": "video", "video": {"video_path": "a.mp4", "fps": 1, "max_frames": 180}},
                                         {"type": "text", "text": "Q1?"}]},
            {"role": "assistant", "content": "A1."},
            {"role": "user", "content": [{"type": "video", "video": {"video_path": "b.mp4", "fps": 1, "max_frames": 180}},
                                         {"type": "text", "text": "Q2?"}]},
            {"role": "assistant", "content": "A2."},
            {"role": "user", "content": [{"type": "video", "video": {"video_path": "c.mp4", "fps": 1, "max_frames": 180,
                                                                     "start_time": 0.0, "end_time": 4.0}},
                                         {"type": "video", "video": {"video_path": "c.mp4", "fps": 1, "max_frames": 180,
                                                                     "start_time": 5.0, "end_time": 9.0}},
                                         {"type": "text", "text": "Q3?"}]}]
    new_conv, all_ts = P.Videollama3Qwen2Processor._load_multimodal_data(me, copy.deepcopy(conv))
    segs = []
    for m in new_conv:
        if isinstance(m["content"], list):
            for c in m["content"]:
                if isinstance(c, dict) and c.get("type") == "video":
                    segs.append({"num_frames": int(c["num_frames"]), "timestamps": [float(t) for t in c["timestamps"]],
                                 "frame_ids": [int(f[0, 0, 0]) for f in c["video"]]})
    with open(os.path.join(HERE, "video_io.json"), "w") as f:
        json.dump({"specs": {k: list(v) for k, v in specs.items()}, "load_video": cases, "conversation": conv,
                   "segments": segs, "all_timestamps": [float(t) for t in all_ts]}, f, indent=1)
    print("video_io.json", len(cases), "load_video cases,", len(segs), "segments")


if __name__ == "__main__":
    if "--only-video-io" in sys.argv:
        golden_video_io()
        sys.exit(0)
    if "--only-preprocess" in sys.argv:
        golden_preprocess()
        sys.exit(0)
    which = sys.argv[1:] or ["preprocess", "vit", "vit_bf16", "kmeans", "kmeans_reseed", "compress", "text", "e2e", "e2e_blockdiag", "qwen2", "lora", "image_modality", "truncate"]
    with torch.no_grad():
        for w in which:
            globals()["golden_" + w]()
