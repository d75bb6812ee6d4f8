"""BASELINE.json `configs` as GPU tests at VideoLLaMA3-7B dimensions (random-init weights: no checkpoint is reachable).

  configs[0]  8 x 224 x 224 clip: processor -> ViT (grid 16 x 16, seq 256) -> projector vs the oracle
  configs[1]  64-frame 480p clip: the whole qa_selection -> generate path with the REAL tokenizer's ids (replayed),
              keep-mask bit-equal to the oracle's at full size, deterministic greedy tokens
  configs[2]  256-frame clip: the 140 x 280 grid (200 patches per frame = one 128-row + one ragged 72-row query block)
              vs the oracle, frame-separable at full length, k-means gate (K = 18) reached; a 2-rank run of the HIP
              encoder (two processes on this one GPU over gloo) equals the single-process encode
  configs[3]  8-turn streaming session: history retrieval from turn 2 on, visual-token cache and prefix-KV reuse
              transparent at real dimensions
(configs[4] = 8 independent replicas needs 8 GPUs; its sharding logic is CPU-tested.)"""
import json
import os
import random
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BF = torch.bfloat16


def _dist(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    d = a - b
    return float(d.abs().max() / b.abs().max()), float(d.pow(2).mean().sqrt() / b.pow(2).mean().sqrt())


@pytest.fixture(scope="module")
def vit2():
    """2-layer ViT at the real width + projector, states shared by the oracle and both HIP precisions"""
    from cogstream_amd.weights import VisionConfig, random_proj_state, random_vit_state
    cfg = VisionConfig(num_hidden_layers=2)
    return cfg, random_vit_state(cfg, seed=13, std=0.03), random_proj_state(cfg.hidden_size, 3584, seed=14, std=0.02)


def _encode_vs_oracle(dev, vit2, frames_u8, n_frames_budget, expect_grid):
    """processor (GPU pre-processing) -> ViT -> projector in bf16 and in the fp32 parity mode vs oracle.vision"""
    from cogstream_amd.preprocess_gpu import preprocess_videos_gpu
    from cogstream_amd.vision import Projector, VisionEncoder
    from oracle import vision as ov
    cfg, st, pst = vit2
    t = frames_u8.shape[0]
    ft = preprocess_videos_gpu([torch.from_numpy(frames_u8).to(dev)], merge_size=2, max_tokens=16384 * t // n_frames_budget,
                               out_dtype=torch.float32)
    grid, merge = ft["grid_sizes"], ft["merge_sizes"]
    assert grid.tolist() == [[t, expect_grid[0], expect_grid[1]]]
    pix = ft["pixel_values"]
    ref_tok = ov.encode(st, pix.cpu(), grid, merge, heads=16, layers=2, mode=0)
    ref = ov.project(pst, ref_tok)
    for dtype, tol in ((BF, 3e-2), (torch.float32, 1e-4)):
        enc = VisionEncoder(st, cfg, dtype=dtype, device=dev)
        proj = Projector(pst, dtype=dtype, device=dev)
        tok = enc(pix.to(dtype), grid, merge)
        out = proj(tok)
        (m1, r1), (m2, r2) = _dist(tok.float(), ref_tok), _dist(out.float(), ref)
        print(f"grid {expect_grid} {dtype}: tokens max {m1:.2e} rms {r1:.2e}; projected max {m2:.2e} rms {r2:.2e}")
        assert m1 < tol and m2 < tol and r1 < tol / 2 and r2 < tol / 2, (dtype, m1, r1, m2, r2)


def test_cfg1_8x224_clip_encoder_vs_oracle(dev, vit2):
    """BASELINE configs[0] on the HIP path: 8 x 224 x 224 -> 16 x 16 patches per frame (seq 256 = two full query blocks)"""
    from cogstream_amd import processing as pr
    frames, _ = pr.synthetic_clip(8, 224, 224, kind="drift", clip_idx=1)
    _encode_vs_oracle(dev, vit2, frames, 8, (16, 16))


def test_cfg3_grid_encoder_vs_oracle_and_frame_separable(dev, vit2):
    """BASELINE configs[2]'s per-frame size: 16 384 tokens over 256 frames -> 140 x 280 = 10 x 20 patches (seq 200)"""
    from cogstream_amd import processing as pr
    from cogstream_amd.preprocess_gpu import preprocess_videos_gpu
    from cogstream_amd.vision import VisionEncoder
    from cogstream_amd.weights import VisionConfig, random_vit_state
    frames, _ = pr.synthetic_clip(12, kind="drift", clip_idx=2)                    # 12 frames of 480 x 854
    _encode_vs_oracle(dev, vit2, frames, 256, (10, 20))
    # the full 256-frame clip through all 27 layers: a 32-frame shard (one GPU's share at N = 8) is bit-identical
    cfg = VisionConfig()
    enc = VisionEncoder(random_vit_state(cfg, 0, dev, BF), cfg, dtype=BF, device=dev)
    clip = np.concatenate([pr.synthetic_clip(64, kind="drift", clip_idx=c)[0] for c in range(4)])
    ft = preprocess_videos_gpu([torch.from_numpy(clip).to(dev)], merge_size=2, max_tokens=16384)
    assert ft["grid_sizes"].tolist() == [[256, 10, 20]] and ft["pixel_values"].shape == (51200, 588)
    merge = torch.tensor([2])
    whole = enc(ft["pixel_values"], ft["grid_sizes"], merge).clone()
    assert whole.shape == (12800, 1152) and bool(torch.isfinite(whole.float()).all())
    for r in (0, 5, 7):
        part = enc(ft["pixel_values"][r * 6400:(r + 1) * 6400], torch.tensor([[32, 10, 20]]), merge)
        assert torch.equal(part, whole[r * 1600:(r + 1) * 1600]), r


@pytest.fixture(scope="module")
def model7b(dev):
    """the full-size model (27-layer ViT, projector, 28-layer Qwen2-7B), random-init on the GPU"""
    from cogstream_amd.chat import CogReasoner
    from cogstream_amd.llm import Qwen2Engine
    from cogstream_amd.vision import Projector, VisionEncoder
    from cogstream_amd.weights import LlmConfig, VisionConfig, random_llm_state, random_proj_state, random_vit_state
    vcfg, lcfg = VisionConfig(), LlmConfig()
    enc = VisionEncoder(random_vit_state(vcfg, 0, dev, BF), vcfg, dtype=BF, device=dev)
    proj = Projector(random_proj_state(1152, 3584, 1, dev, BF), dtype=BF, device=dev)
    eng = Qwen2Engine(random_llm_state(lcfg, 2, dev, BF), lcfg, dtype=BF, device=dev)
    torch.cuda.empty_cache()
    return enc, proj, eng, lcfg


def test_cfg2_full_pipeline_with_real_token_ids(dev, model7b):
    """BASELINE configs[1]: 64-frame 480p clip -> processor (GPU pre-processing, the REAL tokenizer's ids replayed
    from tests/golden/tokenizer.json: 15 395 prompt tokens) -> qa_selection -> generate at full size. The keep-mask
    equals the oracle's on the same pixel_values bit for bit, the compacted prompt has exactly the kept rows, greedy
    decoding is deterministic, and the shipped sampled mode runs on the device sampler."""
    from replay_tokenizer import ReplayTokenizer
    from cogstream_amd import processing as pr
    from cogstream_amd.chat import CogReasoner
    from oracle import compress as oc
    enc, proj, eng, lcfg = model7b
    tok = ReplayTokenizer()
    model = CogReasoner(enc, proj, eng, lcfg)                       # DEFAULT_GENERATION = generation_config.json
    base, ts = pr.synthetic_clip(64, kind="noise", clip_idx=0)
    frames = np.repeat(base[:1], 64, axis=0)                       # static background (no sensor noise) ...
    for i in range(64):                                            # ... and a 160 x 160 textured square moving 8 px per frame
        frames[i, 160:320, 8 * i:8 * i + 160] = base[1, 160:320, :160]
    conv = [{"role": "user", "content": [{"type": "video", "video": frames, "timestamps": ts},
                                         {"type": "text", "text": "What is happening in the video?"}]}]
    inputs = pr.CogStreamProcessor(tok, device=dev)(conversation=conv, add_system_prompt=True, add_generation_prompt=True)
    assert inputs["input_ids"].shape == (1, 15395) and inputs["grid_sizes"].tolist() == [[64, 22, 42]]
    assert int((inputs["input_ids"] == 151665).sum()) == 64 * 231 == 14784
    sel = model.qa_selection(**inputs, mode="FCC")
    assert sel["selection_module_output"] == "" and sel["if_visual"] is True            # no history: no retrieval
    runs = []
    for _ in range(2):
        ids, _ = model.generate(**sel, max_new_tokens=12, do_sample=False, eos_token_id=[-1])
        runs.append(ids[0].tolist())
    assert runs[0] == runs[1] and len(runs[0]) == 12 and all(0 <= t < lcfg.vocab_size for t in runs[0])
    mask = model.last_debug["compression_mask"].cpu()
    want = oc.compression_mask(inputs["pixel_values"].cpu(), inputs["grid_sizes"], inputs["merge_sizes"], ["video"],
                               minor_frame_indices=[])
    assert torch.equal(mask.bool(), want) and 231 + 63 <= int(mask.sum()) < 14784       # frame 0 whole, >= 1 per frame
    assert model.last_debug["minor_frames"] == []                                          # ceil(64/15) = 5 <= 9: no k-means
    assert model.last_debug["input_ids"].numel() == 15395 - (14784 - int(mask.sum()))
    ids, _ = model.generate(**sel, max_new_tokens=12, eos_token_id=[-1], seed=7)          # shipped mode: sampled
    ids2, _ = model.generate(**sel, max_new_tokens=12, eos_token_id=[-1], seed=7)
    assert ids.shape == (1, 12) and ids.tolist() == ids2.tolist()


def test_cfg3_event_selection_reaches_kmeans_at_full_size(dev, model7b):
    """256 frames -> K = ceil(256/15) = 18 clusters, 19-sequence event-summary prefill, event compression; integer
    products are consistent (every frame assigned, minor frames exclude the near-centroid picks, mask rows of minor
    frames collapse to one token)"""
    from toy_tokenizer import IM_END, IMAGE, ToyTokenizer
    from cogstream_amd import processing as pr
    from cogstream_amd.chat import CogReasoner
    from cogstream_amd.weights import LlmConfig
    enc, proj, eng, _ = model7b
    lcfg = LlmConfig(image_token_index=IMAGE, eos_token_id=IM_END)     # byte tokenizer: ids < 512 of the 152064 rows
    model = CogReasoner(enc, proj, eng, lcfg, generation_config=dict(do_sample=False, eos_token_id=[-1], repetition_penalty=1.05))
    tok = ToyTokenizer()
    clip = np.concatenate([pr.synthetic_clip(64, kind="drift", clip_idx=c)[0] for c in range(4)])
    conv = [{"role": "user", "content": [{"type": "video", "video": clip, "timestamps": [float(i) for i in range(256)]},
                                         {"type": "text", "text": "What is happening in the video?"}]}]
    inputs = pr.CogStreamProcessor(tok, device=dev)(conversation=conv, add_system_prompt=True, add_generation_prompt=True)
    assert inputs["grid_sizes"].tolist() == [[256, 10, 20]] and inputs["total_image_num"] == 256
    random.seed(3)
    torch.manual_seed(3)
    sel = model.qa_selection(**inputs, mode="FCC")
    model.cosine_override = [0.9 if k % 3 else 0.1 for k in range(18)]            # random weights: force 6 minor events
    ids, _ = model.generate(**sel, max_new_tokens=4)
    dbg = model.last_debug
    assert len(dbg["assign"]) == 256 and set(dbg["assign"]) <= set(range(18)) and len(dbg["cosine_raw"]) == 18
    minor = dbg["minor_frames"]
    assert minor == sorted(minor) and 0 < len(minor) < 256
    assert all(dbg["assign"][f] % 3 == 0 for f in minor)                            # only frames of the forced events
    mask = dbg["compression_mask"].cpu().view(256, 50)
    assert all(int(mask[f].sum()) == 1 and bool(mask[f, 0]) for f in minor)         # one pooled token per minor frame
    assert ids.shape == (1, 4)
    model.cosine_override = None
    # How close were the k-means decisions on REAL encoder outputs (not planted data)? The product reports the smallest
    # relative margin between the best and the second-best cluster of any row in any iteration. At or above 1e-3 the
    # reference's arithmetic (torch.cdist's sgemm form, oracle.kmeans with its default distances) must give the same
    # assignment; below it the reference's own rounding decides and only the count is recorded.
    from cogstream_amd import kmeans as km
    from oracle import kmeans as ok
    ts = torch.arange(256, dtype=torch.float32)
    for kind in ("drift", "noise"):
        clip_k = clip if kind == "drift" else np.concatenate([pr.synthetic_clip(64, kind="noise", clip_idx=c)[0] for c in range(4)])
        ft = pr.CogStreamProcessor(tok, device=dev).process_images([("video", clip_k)])
        mm = model.encode_images(ft["pixel_values"], ft["grid_sizes"], ft["merge_sizes"])
        feats = mm.view(256, 50, 3584)
        random.seed(11)
        torch.manual_seed(11)
        _, _, assign = km.kmeans_with_time_min_max(feats, ts, 18)
        st = dict(km.last_stats)
        print(f"k-means margins on cfg3 '{kind}' encoder outputs: min relative margin {st['min_rel_margin']:.3e}, "
              f"(row, iteration) pairs below 1e-3: {st['rows_below_1e-3']} of {256 * st['iterations']} ({st['iterations']} iterations), "
              f"k-means++ path: {st['kpp_path']}")
        assert st["kpp_path"] == "one call" and st["min_rel_margin"] >= 0 and st["iterations"] >= 1
        random.seed(11)
        torch.manual_seed(11)
        _, _, want = ok.kmeans_with_time_min_max(feats.float().cpu(), ts, 18)
        same = bool(torch.equal(assign.cpu(), want))
        print(f"    reference arithmetic (oracle, torch.cdist's sgemm form) gives the same 256 assignments: {same}"
              + ("" if same else f" ({int((assign.cpu() != want).sum())} rows differ)"))
        if st["rows_below_1e-3"] == 0:
            assert same, kind


def test_cfg4_eight_turn_session_caches_are_transparent_at_full_size(dev, model7b):
    """BASELINE configs[3]: 8 turns, one new 8-frame 480p segment + question per turn (tests/golden/cfg4_history.json),
    history retrieval from turn 2 on. The session with the visual-token cache + prefix-KV reuse produces the same
    answers and selections as the one without; caches are really used."""
    from toy_tokenizer import IM_END, IMAGE, ToyTokenizer
    from cogstream_amd import processing as pr
    from cogstream_amd.answer_generate import run_session
    from cogstream_amd.chat import CogReasoner
    from cogstream_amd.weights import LlmConfig
    enc, proj, eng, _ = model7b
    lcfg = LlmConfig(image_token_index=IMAGE, eos_token_id=IM_END)
    tok = ToyTokenizer()
    hist = json.load(open(os.path.join(ROOT, "tests", "golden", "cfg4_history.json")))["turns"]
    segs = []
    for i in range(8):
        fr, ts = pr.synthetic_clip(8, kind="drift", clip_idx=i)
        segs.append({"video": fr, "timestamps": [t + 8 * i for t in ts], "questions": [hist[i]["question"]]})
    recs, stats = [], None
    for cached in (False, True):
        model = CogReasoner(enc, proj, eng, lcfg, generation_config=dict(do_sample=False, eos_token_id=[-1], repetition_penalty=1.05))
        if cached:
            model.enable_visual_cache()
            model.enable_prefix_cache()
        random.seed(0)
        torch.manual_seed(0)
        recs.append(run_session(model, pr.CogStreamProcessor(tok, device=dev), segs, max_new_tokens=6))
        if cached:
            stats = (model.visual_cache_stats, model.prefix_cache_stats())
    assert [r["qa_id"] for r in recs[0]] == list(range(8))
    assert recs[0][0]["predicted_coi"] == [] and [len(r["predicted_coi"]) for r in recs[0]] == list(range(8))
    assert [r["prediction"] for r in recs[0]] == [r["prediction"] for r in recs[1]]
    assert [r["predicted_coi"] for r in recs[0]] == [r["predicted_coi"] for r in recs[1]]
    vis, pre = stats
    assert vis["hits"] + vis["misses"] == 36 and vis["hits"] > 0                      # turn t looks up t segments
    assert pre["selection"][0] > 0 and pre["selection"][1] > pre["selection"][0]       # retrieval prompts share a prefix


_RANK_SCRIPT = r'''
import os, sys, torch, numpy as np
import torch.distributed as dist
sys.path.insert(0, os.environ["COGS_ROOT"])
from cogstream_amd import processing as pr
from cogstream_amd.parallel import frame_shards, gather_tokens
from cogstream_amd.preprocess_gpu import preprocess_videos_gpu
from cogstream_amd.vision import Projector, VisionEncoder
from cogstream_amd.weights import VisionConfig, random_proj_state, random_vit_state
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
dev = torch.device("cuda:0")                      # rehearsal: both ranks share the one GPU of this box
cfg = VisionConfig(num_hidden_layers=3)
enc = VisionEncoder(random_vit_state(cfg, 0, dev, torch.bfloat16), cfg, device=dev)
proj = Projector(random_proj_state(1152, 3584, 1, dev, torch.bfloat16), device=dev)
T = 20
clip = pr.synthetic_clip(T, kind="drift", clip_idx=0)[0]
lo, hi = frame_shards(T, world)[rank]
ft = preprocess_videos_gpu([torch.from_numpy(clip[lo:hi]).to(dev)], merge_size=2, max_tokens=16384 * (hi - lo) // 256)
_, gh, gw = (int(v) for v in ft["grid_sizes"][0])
mine = proj(enc(ft["pixel_values"], torch.tensor([[hi - lo, gh, gw]]), torch.tensor([2])))
allt = gather_tokens(mine, (T, gh, gw), 2, world)
if rank == 0:
    fw = preprocess_videos_gpu([torch.from_numpy(clip).to(dev)], merge_size=2, max_tokens=16384 * T // 256)
    whole = proj(enc(fw["pixel_values"], fw["grid_sizes"], torch.tensor([2])))
    ok = bool(torch.equal(whole, allt)) and (gh, gw) == (10, 20) and allt.shape == (T * 50, 3584)
    print("RANK2_RESULT", "OK" if ok else "MISMATCH", tuple(allt.shape), flush=True)
dist.barrier()
dist.destroy_process_group()
'''


def test_two_rank_hip_encoder_equals_single_process(dev, tmp_path):
    """N > 1 rehearsal with the HIP encoder on every rank (not the oracle): two processes on this one GPU, gloo
    rendezvous on 127.0.0.1, 20 frames of the cfg3 grid split 10 + 10, ragged-free all-gather; rank 0 checks the
    gathered [1000, 3584] tokens against the whole-clip encode bit for bit"""
    script = tmp_path / "rank2.py"
    script.write_text(_RANK_SCRIPT)
    env = dict(os.environ, COGS_ROOT=ROOT, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", "29517", str(script)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "RANK2_RESULT OK" in r.stdout, r.stdout[-2000:]


def test_bench_cfg3_mode_and_extra_key(dev):
    """bench.py --config cfg3 times BASELINE configs[2] (256 frames at 140 x 280); the default run carries it as the
    extra key `cfg3` beside the cfg2 headline"""
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "cfg3", "--steps", "2", "--warmup", "1",
                        "--no-llm", "--no-cpu"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["scaling"] == "strong" and d["config"]["frames"] == 256 and d["config"]["patches"] == 51200
    assert "140x280" in d["config"]["workload"] and d["value"] > 0 and "cfg3" not in d
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-llm", "--no-cpu"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["config"]["frames"] == 64 and d["scaling"] == "strong" and d["cfg3"]["frames_per_gpu"] == 256 and d["ranks_seen"] == 1
    assert "140x280" in d["cfg3"]["workload"] and d["cfg3"]["value"] > 0
    assert d["roofline"]["traffic_source"] is None or "not this run" in d["roofline"]["traffic_source"]


def test_reference_driver_call_sequence_under_the_reference_names(dev, tmp_path):
    """evaluate/answer_generate.py:173-183 (load) and :60-76 (infer), statement for statement, with ONE import changed:
    AutoModelForCausalLM / AutoProcessor / PeftModel come from cogstream_amd.auto, which resolves the checkpoint
    directory's auto_map (model/config.json:6-9, model/preprocessor_config.json:2-5) to
    cogstream_amd.cogreasoner_chat.Videollama3Qwen2ForCausalLM / processing_cogreasoner.Videollama3Qwen2Processor.
    The checkpoint is synthesised (real vocabulary size, small widths), the tokenizer is the REAL one replayed
    (BASELINE configs[0] conversation: 621 prompt ids), both adapters are peft directories."""
    from safetensors.torch import save_file
    from replay_tokenizer import ReplayTokenizer
    from cogstream_amd import checkpoint as ck
    from cogstream_amd import processing as pr
    from cogstream_amd.auto import AutoModelForCausalLM, AutoProcessor, PeftModel        # <- the one changed import
    from cogstream_amd.cogreasoner_chat import Videollama3Qwen2ForCausalLM
    from cogstream_amd.processing_cogreasoner import Videollama3Qwen2Processor
    from cogstream_amd.weights import (LlmConfig, VisionConfig, random_llm_state, random_lora_state, random_proj_state,
                                       random_vit_state)

    class Tok(ReplayTokenizer):
        def decode(self, ids, skip_special_tokens=False):          # generated ids are random: no recorded text for them
            ids = ids.tolist() if hasattr(ids, "tolist") else list(ids)
            return self.by_ids.get(tuple(ids), " ".join(str(i) for i in ids))

    vcfg = VisionConfig(hidden_size=576, intermediate_size=200, num_hidden_layers=2, num_attention_heads=8)
    lcfg = LlmConfig(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=1)
    model_path = str(tmp_path / "ckpt")
    ck.save_checkpoint(model_path, random_vit_state(vcfg, seed=3, std=0.05),
                       random_proj_state(vcfg.hidden_size, lcfg.hidden_size, seed=1, std=0.05),
                       random_llm_state(lcfg, seed=7, std=0.05), vcfg, lcfg,
                       generation={"do_sample": False, "eos_token_id": [151645, 151643]}, n_shards=2)
    adapters = []
    for i, name in enumerate(("full", "language")):
        ad = str(tmp_path / f"adapter_{name}")
        os.makedirs(ad)
        lora = random_lora_state(lcfg, seed=20 + i, r=8, proj_dims=(vcfg.hidden_size, lcfg.hidden_size))
        save_file({k: v.contiguous() for k, v in lora.items()}, os.path.join(ad, "adapter_model.safetensors"))
        json.dump({"r": 8, "lora_alpha": 16}, open(os.path.join(ad, "adapter_config.json"), "w"))
        adapters.append(ad)
    local_rank = dev.index or 0
    torch.cuda.set_device(local_rank)

    # ---- evaluate/answer_generate.py:173-183
    model = AutoModelForCausalLM.from_pretrained(
        model_path,
        trust_remote_code=True,
        torch_dtype=torch.bfloat16,
        attn_implementation="flash_attention_2",
    )
    processor = AutoProcessor.from_pretrained(model_path, trust_remote_code=True, tokenizer=Tok())
    tokenizer = processor.tokenizer
    model = PeftModel.from_pretrained(model, adapters[0], adapter_name="full_module")
    model.load_adapter(adapters[1], adapter_name="language_module")
    model.to(local_rank)
    assert type(model) is Videollama3Qwen2ForCausalLM and type(processor) is Videollama3Qwen2Processor
    assert tokenizer is processor.tokenizer and model.active_adapter == "full_module"

    # ---- evaluate/answer_generate.py:60-76 (infer), on the BASELINE configs[0] conversation
    frames, ts = pr.synthetic_clip(8, 224, 224)
    conversation = [{"role": "user", "content": [{"type": "video", "video": frames, "timestamps": ts},
                                                 {"type": "text", "text": "What is happening in the video?"}]}]
    select, if_visual = None, None
    inputs = processor(
        conversation=conversation,
        add_system_prompt=True,
        add_generation_prompt=True,
        return_tensors="pt"
    )
    assert inputs["input_ids"].shape == (1, 621)
    inputs = {k: v.to(model.device) if isinstance(v, torch.Tensor) else v for k, v in inputs.items()}
    if "pixel_values" in inputs:
        inputs["pixel_values"] = inputs["pixel_values"].to(dtype=torch.bfloat16)
    model.set_adapter("language_module")
    inputs = model.qa_selection(**inputs, mode="FCC", select_gt=select, if_visual=if_visual)
    model.set_adapter("full_module")
    output_ids, selection_module_output = model.generate(**inputs, max_new_tokens=6)
    response = processor.batch_decode(output_ids, skip_special_tokens=True)[0].strip()
    assert isinstance(response, str) and selection_module_output == "" and output_ids.shape[0] == 1
    assert 1 <= output_ids.shape[1] <= 6
    # the adapters are live: the answer stage under the other adapter / the base weights decodes differently
    lg = {}
    emb = model.llm.embed_tokens(torch.arange(100, 164, device=dev))
    for name in ("full_module", "language_module", "base"):
        model.set_adapter(name)
        lg[name] = model.llm.forward(emb)["logits"].clone()
    assert rel_err(lg["full_module"], lg["base"]) > 1e-3 and rel_err(lg["language_module"], lg["full_module"]) > 1e-3


def test_c_abi_allgather_over_rccl_single_rank(dev):
    """cogs_allgather_tokens (the C-ABI form of the path's one collective) on a real RCCL
    communicator: this box has one GPU, so the communicator has one rank -- what is checked is the binding (librccl is
    loaded lazily, the call lands on ncclAllGather with a byte payload on the caller's stream) and the argument
    checks. The N > 1 data movement is RCCL's own; the frame-order bookkeeping around it is covered by the gloo tests
    (tests/test_distributed_cpu.py) and the two-rank HIP rehearsal above."""
    import ctypes as C
    from cogstream_amd import _lib as L
    try:
        rccl = C.CDLL("librccl.so.1")
    except OSError:
        pytest.skip("librccl not present")

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]

    uid = UniqueId()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    try:
        local = torch.randn(1600, 3584, device=dev).to(BF)                 # one rank's cfg3 share: 32 frames x 50 tokens
        out = torch.zeros_like(local)
        rc = L.lib.cogs_allgather_tokens(L.current_stream(), comm, local.data_ptr(), local.shape[0], local.shape[1] * 2,
                                         out.data_ptr())
        torch.cuda.synchronize()
        assert rc == L.OK and torch.equal(out, local)
        assert L.lib.cogs_allgather_tokens(L.current_stream(), None, local.data_ptr(), 1600, 7168, out.data_ptr()) == L.E_INVALID
        assert L.lib.cogs_allgather_tokens(L.current_stream(), comm, local.data_ptr(), 0, 7168, out.data_ptr()) == L.E_INVALID
    finally:
        rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        rccl.ncclCommDestroy(comm)
