#!/usr/bin/env python3
"""bench.py -- the reference's headline metric on MI355X: frames/sec encoded (+ answer tokens/sec),
64-frame 480p clip, VideoLLaMA3-7B dimensions, bf16 (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W

N > 1 as a plain command: the parent process -- before it touches the GPU -- starts N fresh rank processes
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same flags>`),
relays rank 0's JSON line and exits with their status. Started under torch.distributed.run by someone else
(RANK / WORLD_SIZE in the environment) it is simply one of the ranks.

One STEP = one pass of the encoder stage over the clip with pixel_values already resident in HBM:
cogs_vit_encode (patch-embed GEMM, 27 x {LN, QKV GEMM+RoPE, per-frame attention, out-proj, LN, MLP},
post-LN + 2x2 merge) + cogs_project -- SURVEY.md section 8(d) "frames/sec encoded = T / t(A1..A7)".
N > 1: the clip's frames are sharded, rank r encodes and projects its contiguous slice, ONE RCCL all-gather
reassembles the [M,3584] visual tokens in frame order on every rank (the LLM rank needs them all); time = max
over ranks. The headline at EVERY N is the metric's own clip -- BASELINE configs[1], 64 frames -- so the N = 1, 2, 4, 8
values are one STRONG-scaling curve of a fixed workload (32 / 16 / 8 frames per GPU). Side keys of the same line:
  cfg3   BASELINE configs[2]: ONE 256-frame 480p clip (16 384-token budget -> 140x280 per frame, 200 patches / 50
         tokens per frame), frames sharded over the N ranks (32 per GPU at N = 8), one all-gather -- strong scaling
  weak   N > 1: the clip grows with N at 64 frames per GPU (every rank keeps the N = 1 workload; the only added cost is
         the all-gather) -- near-linear by construction, reported for completeness
`--config cfg3` makes configs[2] the timed headline instead; `--scaling weak` the weak curve; `--config cfg5` is
BASELINE configs[4]: every rank answers its OWN 64-frame clip through the whole product API (processor -> qa_selection
-> generate), replicas only, no collective.

The same JSON line also carries, measured after the timed steps on rank 0 at N = 1:
  answer_tokens_per_s   greedy decode rate of the Qwen2-7B path (prefill of the full ~15k-token
                        interleaved prompt, then 128 tokens, EOS ignored), e2e_s the whole answer latency;
  pipeline              wall time of CogStreamProcessor -> qa_selection -> generate (64 new tokens) through the product
                        API at cfg2 and cfg3 with real-tokenizer prompt lengths, split by stage;
  roofline              the dominant kernel (the bf16 MFMA GEMM): algorithmic FLOPs of the encoder's GEMM
                        launches / their summed duration, measured live with HIP events on the launch stream;
  cpu_baseline          the oracle (CPU fp32 restatement of the reference) timed on this box's host cores
                        on a bounded sample of the same workload (rank 0, N = 1 only).
Weights are random-init at the real dimensions (no checkpoint is reachable); data is synthetic."""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md
BARE_MFMA_TFLOPS = 2050.0     # measured, tools/micro/mfma_shape_dvfs.cpp on MI355X: 16x16x32 bf16, random operands, 1-2 waves per SIMD


def vit_gemm_flops(n_patches: int, m_tokens: int, cfg, llm_hidden: int) -> float:
    """algorithmic (unpadded) GEMM FLOPs of one encode + project"""
    H, I, L, pd = cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers, cfg.patch_dim
    return (2.0 * n_patches * L * (4 * H * H + 2 * H * I) + 2.0 * n_patches * pd * H
            + 2.0 * m_tokens * (H * llm_hidden + llm_hidden * llm_hidden))


def vit_attn_flops(frames: int, per_frame: int, cfg) -> float:
    return 4.0 * cfg.num_hidden_layers * frames * per_frame * per_frame * cfg.hidden_size


def frame_header_tokens(t: int) -> int:
    """'Time <t>.0s:' with the real Qwen2 tokenizer: 'Time', ' ', one token per digit, '.', '0', 's', ':'
    ('Time 12.0s:' -> 8 tokens; tests/golden/tokenizer.json)"""
    return 6 + len(str(int(t)))


def prompt_tokens(T: int, P: int) -> int:
    """length of the answer prompt the processor builds for a T-frame clip at 1 fps with P visual tokens per frame
    and the bench question (chat template, processing_cogreasoner.py:707-730,752-801): default system turn (30) +
    '<|im_start|>user\n' (3) + per frame header + P + separator (1) + question (7) + '<|im_end|>\n' (2) +
    generation prompt (3). Pinned against the real tokenizer: 15395 at cfg2 (tests/test_tokenizer_golden.py)."""
    return 30 + 3 + sum(frame_header_tokens(t) + P + 1 for t in range(T)) + 7 + 2 + 3


HBM_PEAK_TBPS = 8.0        # HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md (6.29 TB/s is the measured copy ceiling)


def hbm_stage(name, kernel, ref, nbytes, seconds, note, cache_seconds=None):
    """one HBM-bound stage of SURVEY.md section 8(d): algorithmic bytes / wall time of the stage vs the 8 TB/s peak.
    `seconds` is measured on ROTATING buffers whose total exceeds the 256 MiB Infinity Cache (every call finds its
    operands in HBM); cache_seconds, when given, is the same call repeated on ONE set of buffers (operands resident in
    the Infinity Cache after the first call) -- reported beside it as `cache_resident`, never as the roofline figure."""
    tbps = nbytes / seconds / 1e12
    rec = {"stage": name, "kernel": kernel, "reference": ref, "bound": "hbm", "algorithmic_bytes": int(nbytes),
           "ms": round(seconds * 1e3, 4), "achieved_TBps": round(tbps, 3), "peak_TBps": HBM_PEAK_TBPS,
           "frac": round(tbps / HBM_PEAK_TBPS, 4), "note": note}
    if cache_seconds is not None:
        rec["cache_resident"] = {"ms": round(cache_seconds * 1e3, 4), "achieved_TBps": round(nbytes / cache_seconds / 1e12, 3),
                                 "note": "same call back to back on ONE set of buffers: served by the 256 MiB Infinity "
                                         "Cache, not HBM (the round-3/4 figure)"}
    return rec


def stage_rooflines(c3, mm3, dev, with_cpu=True):
    """The HBM-bound stages of the path at BASELINE configs[2] size (256 frames, 50 tokens per frame), each timed as
    the product calls it (host control included -- k-means keeps the reference's host RNG draws):
    k-means (A8), pixel-difference mask (A12), compaction + embedding splice (A13-A14)."""
    import random
    from cogstream_amd import kmeans as km
    from cogstream_amd import ops

    def timed(fn, n=5, reps=1):
        """median over n samples of (reps back-to-back calls + one synchronise) / reps: with reps > 1 the per-call
        launch + synchronise latency of a 10 us kernel does not pass for its duration"""
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            for _ in range(reps):
                r = fn()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / reps)
        return r, sorted(ts)[len(ts) // 2]

    T3 = 256
    P3 = mm3.shape[0] // T3
    D = mm3.shape[1]
    es = mm3.element_size()
    feats = mm3.view(T3, P3, D)
    tsd = torch.arange(T3, dtype=torch.float32)
    K = -(-T3 // 15)

    def run_km():
        random.seed(0)
        torch.manual_seed(0)
        return km.kmeans_with_time_min_max(feats, tsd, K)

    _, t_km = timed(run_km, 3)
    st = dict(km.last_stats)
    passes = st["kpp_passes"] + 2 * st["iterations"]
    out = [hbm_stage("kmeans_with_time_min_max (A8)", "sqdist_kernel / update_kernel (csrc/kmeans.hip)",
                     "model/kmeans_with_time.py:4-137", passes * T3 * P3 * D * es, t_km,
                     f"[{T3},{P3 * D}] bf16 features, K={K}: {st['kpp_passes']} k-means++ passes + {st['iterations']} Lloyd "
                     f"iterations x 2 passes (distances, means) over the features; wall time of the whole call (seeding in one "
                     f"library call from host-drawn exponentials, Lloyd iterations queued four at a time; {st.get('kpp_path', '')}). "
                     f"The algorithm itself re-reads the SAME {T3 * P3 * D * es / 1e6:.0f} MB of features on every pass, so all "
                     f"passes after the first are Infinity-Cache reads by construction -- not an artefact of the timing loop",
                     )]
    out[0]["min_rel_margin"], out[0]["rows_below_1e-3"] = st.get("min_rel_margin"), st.get("rows_below_1e-3")
    pix3 = c3["pix"]
    gh3, gw3 = c3["gh"], c3["gw"]
    Pm = gh3 * gw3 // 4
    minor = torch.zeros(T3, dtype=torch.uint8, device=dev)
    _, t_pd_cache = timed(lambda: ops.pixdiff_mask(pix3, T3, Pm, 0.1, 1, minor), 5, reps=50)
    # rotating operands: 6 copies of the 60 MB pixel_values = 361 MB > the 256 MiB Infinity Cache, visited round robin
    pix_ring = [pix3] + [pix3.clone() for _ in range(5)]
    ring_i = [0]

    def pd_rot():
        ring_i[0] = (ring_i[0] + 1) % len(pix_ring)
        return ops.pixdiff_mask(pix_ring[ring_i[0]], T3, Pm, 0.1, 1, minor)

    _, t_pd = timed(pd_rot, 5, reps=48)
    del pix_ring
    out.append(hbm_stage("_get_compression_mask (A12)", "pixdiff_kernel (csrc/compress.hip)",
                         "model/cogreasoner_chat.py:383-432", pix3.numel() * pix3.element_size(), t_pd,
                         f"pixel_values [{pix3.shape[0]},588] bf16 read once, uint8 mask [{T3 * Pm}] written; two launches "
                         f"(difference + per-frame fix-up), 48 calls back to back per sample over 6 rotating copies of the "
                         f"input (361 MB in all)", cache_seconds=t_pd_cache))
    # compaction + splice: every prompt row is one gathered row (embedding table or visual token), cogs_gather_rows
    S = mm3.shape[0] + 2048
    table = torch.randn(4096, D, device=dev, dtype=mm3.dtype)
    idx = torch.cat([torch.arange(2048, device=dev), -torch.arange(1, mm3.shape[0] + 1, device=dev)]).to(torch.int64)
    _, t_g_cache = timed(lambda: ops.gather_rows(table, mm3, idx), 5, reps=50)
    # rotating operands: 3 copies of the visual tokens (92 MB each) and 3 outputs (106 MB each) = 594 MB
    mm_ring = [mm3] + [mm3.clone() for _ in range(2)]
    out_ring = [torch.empty(S, D, device=dev, dtype=mm3.dtype) for _ in range(3)]

    def g_rot():
        ring_i[0] = (ring_i[0] + 1) % 3
        return ops.gather_rows(table, mm_ring[ring_i[0]], idx, out=out_ring[ring_i[0]])

    _, t_g = timed(g_rot, 5, reps=48)
    del mm_ring, out_ring
    out.append(hbm_stage("_compress_visual_tokens + prepare_inputs_labels_for_multimodal (A13-A14)",
                         "gather_rows_kernel (csrc/compress.hip)", "model/cogreasoner_chat.py:449-476,567-572",
                         2 * S * D * es, t_g, f"{S} prompt rows of {D} bf16 read + written once; 48 calls back to back per sample "
                         f"over 3 rotating (source, destination) pairs (594 MB in all)", cache_seconds=t_g_cache))
    if with_cpu:
        # the CPU restatement (oracle/, torch fp32 on this box's host cores) of the same stages on the same inputs, once each
        from oracle import compress as oc
        from oracle import kmeans as ok
        ncpu = min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
        torch.set_num_threads(ncpu)
        f_cpu = feats.float().cpu()
        random.seed(0)
        torch.manual_seed(0)
        t0 = time.perf_counter()
        ok.kmeans_with_time_min_max(f_cpu, tsd, K)
        t_km_cpu = time.perf_counter() - t0
        p_cpu = pix3.float().cpu()
        grid3 = torch.tensor([[T3, gh3, gw3]])
        t0 = time.perf_counter()
        oc.compression_mask(p_cpu, grid3, torch.tensor([2]), ["video"], minor_frame_indices=[])
        t_pd_cpu = time.perf_counter() - t0
        tab_cpu, mm_cpu, idx_cpu = table.float().cpu(), mm3.float().cpu(), idx.cpu()
        t0 = time.perf_counter()
        torch.where((idx_cpu >= 0)[:, None], tab_cpu[idx_cpu.clamp_min(0)], mm_cpu[(-idx_cpu - 1).clamp_min(0)])
        t_g_cpu = time.perf_counter() - t0
        for st_, tc in zip(out, (t_km_cpu, t_pd_cpu, t_g_cpu)):
            st_["cpu_baseline"] = {"ms": round(tc * 1e3, 2), "cores": ncpu, "kind": "port",
                                   "speedup": round(tc / (st_["ms"] * 1e-3), 1)}
    return out


class PromptLengthTokenizer:
    """A tokenizer whose token COUNTS on this bench's prompts are the real Qwen2 tokenizer's (the vocabulary does not
    ship: reference data). Qwen2's pre-tokenisation regex cuts the text; a pre-token is one id, except the few words of
    the fixed prompts that the real BPE cuts further (EXTRA, read off the real tokenizer in the build container). The
    ids are arbitrary stand-ins below the special range -- random-init weights do not care; the chat specials and
    <image> have their real ids. Checked against the recorded real lengths: cfg2 15 395, cfg1 621, cfg3 15 295 prompt
    tokens, event-summary prompt 974 (tests/test_tokenizer_golden.py, tests/golden/tokenizer.json)."""
    PRETOKENIZE = (r"(?i:'s|'t|'re|'ve|'m|'ll|'d)|[^\r\n\p{L}\p{N}]?\p{L}+|\p{N}| ?[^\s\p{L}\p{N}]+[\r\n]*|\s*[\r\n]+|\s+(?!\S)|\s+")
    SPECIAL = {"<|im_start|>": 151644, "<|im_end|>": 151645, "<|endoftext|>": 151643, "<image>": 151665}
    EXTRA = {"VideoLLaMA": 3, "DAMO": 1, "summarizing": 1, "timestamped": 1, "Concisely": 2, "adhering": 1}
    GLUED = {"(s"}      # punctuation + letters that ARE one vocabulary entry; any other such pre-token (",Time") is two ids
    init_kwargs = {}
    pad_token_id = 151643

    def __init__(self):
        import regex
        self._split = regex.compile("(" + "|".join(regex.escape(t) for t in self.SPECIAL) + ")")
        self._pre = regex.compile(self.PRETOKENIZE)

    def encode(self, text, add_special_tokens=False):
        import zlib
        out = []
        for part in self._split.split(text):
            if part in self.SPECIAL:
                out.append(self.SPECIAL[part])
            elif part:
                for m in self._pre.finditer(part):
                    w = m.group(0)
                    h = zlib.crc32(w.encode("utf-8"))
                    n = 1 + self.EXTRA.get(w.strip(), 0)
                    if len(w) > 1 and not (w[0].isalpha() or w[0].isspace() or w[0].isdigit()) and w[1].isalpha() and w not in self.GLUED:
                        n += 1
                    out.extend((h + 7919 * i) % 151000 for i in range(n))
        return out

    def __call__(self, text, return_tensors="pt", padding=False, truncation=False, max_length=None, **kw):
        ids = self.encode(text)
        if truncation and max_length is not None:
            ids = ids[:max_length]
        t = torch.tensor([ids], dtype=torch.long)
        return {"input_ids": t, "attention_mask": torch.ones_like(t)}

    def decode(self, ids, skip_special_tokens=False):
        return " ".join(str(int(i)) for i in ids)

    def batch_decode(self, batch, skip_special_tokens=False):
        return [self.decode(b, skip_special_tokens) for b in batch]


QUESTION = "What is happening in the video?"


def pipeline_once(model, processor, dframes, new_tokens, expect_prompt=None):
    """one answer through the product API, as evaluate/answer_generate.py:60-76 drives the reference: processor ->
    qa_selection(mode="FCC") -> generate (greedy, EOS ignored, `new_tokens` tokens). -> stage split in seconds"""
    import random
    random.seed(0)
    torch.manual_seed(0)
    T = dframes.shape[0]
    conversation = [{"role": "user", "content": [{"type": "video", "video": dframes, "timestamps": [float(i) for i in range(T)]},
                                                 {"type": "text", "text": QUESTION}]}]
    model.stage_times = st = {}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    inputs = processor(conversation=conversation, add_system_prompt=True, add_generation_prompt=True, return_tensors="pt")
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    if expect_prompt is not None and int(inputs["input_ids"].shape[1]) != expect_prompt:
        raise RuntimeError(f"prompt of {int(inputs['input_ids'].shape[1])} tokens, the real tokenizer makes {expect_prompt}")
    inputs = model.qa_selection(**inputs, mode="FCC", select_gt=None, if_visual=None)
    ids, _ = model.generate(**inputs, max_new_tokens=new_tokens, do_sample=False, repetition_penalty=1.05, eos_token_id=[])
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    model.stage_times = None
    assert ids.shape == (1, new_tokens)
    split = {"processor": t1 - t0, **{k: st.get(k, 0.0) for k in ("encode", "kmeans", "event_prefill", "answer_prefill", "decode")}}
    split["host_and_small_kernels"] = (t2 - t0) - sum(split.values())
    return {"total_s": round(t2 - t0, 4), "prompt_tokens": int(inputs["input_ids"].shape[1]),
            "kept_visual_tokens": int(model.last_debug["compression_mask"].sum()),
            "minor_frames": len(model.last_debug.get("minor_frames", [])), "event_tokens": model.last_debug.get("event_tokens", 0),
            "new_tokens": new_tokens, "stages_s": {k: round(v, 4) for k, v in split.items()}}


def session_probe(model_factory, processor, dev, turns=8, frames_per_turn=8, new_tokens=32):
    """BASELINE configs[3]: a multi-turn streaming session -- every turn appends one new 8-frame 480p segment and one
    question to the conversation and answers it through the product API exactly as the reference's loop does
    (evaluate/answer_generate.py:130-148 rebuilds the whole conversation each turn; infer = :60-76): historic-dialogue
    retrieval (qa_selection, mode FCC) over the growing history, then generate. Run twice on fresh model objects:
    caches off (the reference's behaviour: everything re-encoded and re-prefilled every turn) and on (visual-token cache
    + prefix-KV reuse). Greedy, `new_tokens` tokens per answer, EOS ignored, prompt lengths of the
    real tokenizer. -> per-turn latencies in seconds."""
    import random
    from cogstream_amd import processing
    from cogstream_amd.answer_generate import infer
    hist = json.load(open(os.path.join(ROOT, "tests", "golden", "cfg4_history.json")))["turns"]
    segs = []
    for i in range(turns):
        fr, ts = processing.synthetic_clip(frames_per_turn, kind="drift", clip_idx=i)
        segs.append((torch.from_numpy(fr).to(dev), [t + frames_per_turn * i for t in ts], hist[i % len(hist)]["question"]))
    rec = {"workload": f"BASELINE configs[3]: {turns} turns, each adds a {frames_per_turn}x480x854 'drift' segment + one question "
                       f"({turns * frames_per_turn} frames by the last turn), FCC history retrieval + answer of {new_tokens} new "
                       f"tokens per turn (greedy), through processor -> qa_selection -> generate",
           "turns": turns, "frames_per_turn": frames_per_turn, "new_tokens": new_tokens}
    for name, cached in (("caches_off", False), ("caches_on", True)):
        model = model_factory()
        if cached:
            model.enable_visual_cache()
            model.enable_prefix_cache()
        lat = []
        for rep in range(2):            # first pass warms allocations and code objects at every prompt size; second is timed
            if cached and rep == 1:
                model = model_factory()
                model.enable_visual_cache()
                model.enable_prefix_cache()
            random.seed(0)
            torch.manual_seed(0)
            conv = [{"role": "system", "content": "You are a helpful assistant."}]
            lat = []
            for fr, ts, q in segs:
                conv.append({"role": "user", "content": [{"type": "video", "video": fr, "timestamps": ts}, {"type": "text", "text": q}]})
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                ans, _ = infer(conv, model, processor, max_new_tokens=new_tokens, do_sample=False, repetition_penalty=1.05,
                               eos_token_id=[])
                torch.cuda.synchronize()
                lat.append(time.perf_counter() - t0)
                conv.append({"role": "assistant", "content": ans})
        rec[name] = {"per_turn_s": [round(x, 4) for x in lat], "session_s": round(sum(lat), 3)}
        if cached:
            rec[name]["visual_cache"] = str(model.visual_cache_stats)
            rec[name]["prefix_rows_reused_of_seen"] = str(model.prefix_cache_stats())
        del model
    rec["speedup_with_caches"] = round(rec["caches_off"]["session_s"] / rec["caches_on"]["session_s"], 3)
    return rec


def build_model(dev, enc=None, proj=None):
    """the full-size model on `dev`, random-init (no checkpoint is reachable), behind the product classes"""
    from cogstream_amd import processing
    from cogstream_amd.chat import CogReasoner
    from cogstream_amd.llm import Qwen2Engine
    from cogstream_amd.vision import Projector, VisionEncoder
    from cogstream_amd.weights import LlmConfig, VisionConfig, random_llm_state, random_proj_state, random_vit_state
    vcfg, lcfg = VisionConfig(), LlmConfig()
    if enc is None:
        enc = VisionEncoder(random_vit_state(vcfg, seed=0, device=dev, dtype=torch.bfloat16), vcfg, dtype=torch.bfloat16, device=dev)
        proj = Projector(random_proj_state(vcfg.hidden_size, lcfg.hidden_size, seed=1, device=dev, dtype=torch.bfloat16),
                         dtype=torch.bfloat16, device=dev)
    eng = Qwen2Engine(random_llm_state(lcfg, seed=2, device=dev, dtype=torch.bfloat16), lcfg, dtype=torch.bfloat16, device=dev)
    torch.cuda.empty_cache()
    model = CogReasoner(enc, proj, eng, lcfg, generation_config=dict(do_sample=False, eos_token_id=[]))
    processor = processing.CogStreamProcessor(PromptLengthTokenizer(), device=dev)
    return model, processor, eng


def bench_cfg5(args, rank, world, dev, ranks_seen):
    """BASELINE configs[4]: one 64-frame 480p clip per GPU, every rank an independent replica of the whole model
    answering its own clip through the product API -- the reference's DistributedSampler mode
    (evaluate/answer_generate.py:186-187). Replicas only: no data-path collective; a step = one answer per rank."""
    import torch.distributed as dist
    from cogstream_amd import processing
    model, processor, _ = build_model(dev)
    fr, _ = processing.synthetic_clip(args.frames, kind=args.clip, clip_idx=rank)
    dfr = torch.from_numpy(fr).to(dev)
    new_tokens = min(args.decode_tokens, 64)
    for _ in range(max(args.warmup, 1)):
        pipeline_once(model, processor, dfr, new_tokens)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        rec = pipeline_once(model, processor, dfr, new_tokens)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device="cpu" if dist.get_backend() == "gloo" else dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt)
    answers = world * args.steps
    return {"metric": "frames/sec encoded + answer tokens/sec, 64-frame clip, VideoLLaMA3-7B",
            "value": round(answers * args.frames / dt, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic", "ranks_seen": ranks_seen,
            "answers_per_s": round(answers / dt, 3), "answer_tokens_per_s_end_to_end": round(answers * new_tokens / dt, 2),
            "config": {"workload": f"cfg5 (BASELINE configs[4]): {world} concurrent {args.frames}x480x854 '{args.clip}' clips, one per GPU, each "
                                   f"answered end to end (processor -> qa_selection -> generate, {new_tokens} new tokens, greedy); value = "
                                   f"frames of the answered clips per second of wall time; random-init weights",
                       "frames": args.frames * world, "frames_per_gpu": args.frames,
                       "parallelism": "replicas only: every rank holds the whole model, no collective (evaluate/answer_generate.py:186-187)"},
            "pipeline_rank0": rec}


def launch_ranks(n: int) -> int:
    """`python bench.py --gpus N` as a plain command: start N fresh rank processes from THIS process, which has not
    touched the GPU (importing torch initialises nothing), relay rank 0's JSON line, return the children's status.
    Never an exec of a GPU-initialised process."""
    import socket
    import subprocess
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    # HSA_ENABLE_IPC_MODE_LEGACY=0: the ROCm runtime of this image shares device memory between processes either through
    # the legacy KFD IPC handles or through dmabuf file descriptors; the host driver of the MI355X boxes supports only
    # the dmabuf form, and with the legacy mode (the runtime's default) RCCL's intra-node transport setup -- every rank
    # maps its peers' buffers with hipIpcGetMemHandle / hipIpcOpenMemHandle -- fails with "hipIpcGetMemHandle: invalid
    # argument". The image exports the variable already; it is set here only so that ranks started from a scrubbed
    # environment still get it. An explicit value in the caller's environment wins.
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    for line in proc.stdout:        # rank 0 prints the one JSON line; anything else the ranks wrote to stdout goes to stderr
        (sys.stdout if line.startswith("{") else sys.stderr).write(line)
        sys.stdout.flush()
    return proc.wait()


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--decode-tokens", type=int, default=128)
    ap.add_argument("--no-llm", action="store_true", help="skip the Qwen2 prefill/decode section")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline")
    ap.add_argument("--no-pipeline", action="store_true", help="skip the product-API pipeline timing")
    ap.add_argument("--clip", default="noise", choices=["noise", "drift"])
    ap.add_argument("--scaling", default="strong", choices=["weak", "strong"],
                    help="N > 1 headline: strong = --frames in total (the metric's clip at every N; default), weak = --frames "
                         "per GPU (clip of frames*N)")
    ap.add_argument("--config", default="cfg2", choices=["cfg2", "cfg3", "cfg5"],
                    help="cfg2: BASELINE configs[1] (64-frame clip; the metric's config). cfg3: configs[2], one 256-frame "
                         "clip at the 16384-token budget (140x280 per frame) sharded over the ranks. cfg5: configs[4], one "
                         "64-frame clip per GPU answered through the whole product API (replicas, no collective)")
    ap.add_argument("--no-cfg3", action="store_true", help="skip the extra configs[2] measurement of the default run")
    ap.add_argument("--payload", default="projected", choices=["projected", "encoder"],
                    help="N > 1: what the all-gather carries (3584-wide projected tokens / 1152-wide encoder tokens, "
                         "projector then runs on every rank)")
    ap.add_argument("--no-session", action="store_true", help="skip the configs[3] multi-turn session timing")
    ap.add_argument("--debug", default="", help="library debug switches for A/B runs: name=value[,name=value...] "
                                                "(cogs_debug_set; `python -c 'from cogstream_amd import _lib; print(_lib.debug_list())'`)")
    ap.add_argument("--vit-streams", type=int, default=2, choices=[1, 2, 3, 4],
                    help="1: encode every clip on one stream (per-kernel profiles); 2 (default): two frame halves on two streams; "
                         "3 / 4: as many contiguous frame ranges (cogs_vit_set_streams; profiles/r5_vit_streams_1to4.txt)")
    ap.add_argument("--no-ln-fold", action="store_true", help="A/B: LayerNorm as its own kernel instead of folded into the GEMMs")
    ap.add_argument("--emulate-shard", type=int, default=8,
                    help="N = 1 only: also time ONE rank's share (1/R of the frames) of the clip on this GPU and report "
                         "shard_efficiency = (t_clip / R) / t_shard (0 = off)")
    args = ap.parse_args()
    t_process0 = time.perf_counter()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch.distributed as dist

    # rehearsal on a one-GPU box only: COGS_BENCH_REHEARSAL=1 puts every rank on cuda:0 and uses gloo (the
    # all-gather then stages through host memory, see parallel.gather_rows); never set by the driver
    rehearsal = os.environ.get("COGS_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    ranks_seen = 1
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
        one = torch.ones(1, device="cpu" if rehearsal else dev)
        dist.all_reduce(one)                 # counted through the communicator, not read from the environment
        ranks_seen = int(one.item())

    from cogstream_amd import _lib as L
    from cogstream_amd import processing
    from cogstream_amd.vision import Projector, VisionEncoder
    from cogstream_amd.weights import (LlmConfig, VisionConfig, random_llm_state, random_proj_state,
                                       random_vit_state)

    vcfg, lcfg = VisionConfig(), LlmConfig()
    debug_applied = L.debug_from_spec(args.debug)
    from cogstream_amd.parallel import frame_shards, gather_tokens
    from cogstream_amd.preprocess_gpu import preprocess_videos_gpu
    if args.config == "cfg5":
        out = bench_cfg5(args, rank, world, dev, ranks_seen)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps(out), flush=True)
        return
    cfg3 = args.config == "cfg3"
    if cfg3:
        args.frames, args.scaling = 256, "strong"
    weak = world > 1 and args.scaling == "weak"
    T = args.frames * world if weak else args.frames          # frames of the whole clip
    torch.zeros(1).to(dev)    # context / allocator warm-up, so that h2d_ms is the copy
    torch.cuda.synchronize()

    def make_clip(T_clip: int, budget_frames: int):
        """this rank's frames of a T_clip-frame synthetic clip -> pixel_values on the GPU, outside the timed region.
        Each rank makes only its own frames (clip content is per 64-frame chunk, so the N = 1 cfg2 clip is chunk 0)
        and pre-processes them on its GPU. Token budget: the processor's 16384 visual tokens per `budget_frames`
        frames (64 -> 308x588 per frame = cfg2's size at any clip length; 256 -> 140x280 = cfg3), scaled with the
        shard so that every rank picks the same per-frame size."""
        lo, hi = frame_shards(T_clip, world)[rank]
        chunks = sorted({f // 64 for f in range(lo, hi)})
        parts = {c: processing.synthetic_clip(64, kind=args.clip, clip_idx=c)[0] for c in chunks}
        fr = np.stack([parts[f // 64][f % 64] for f in range(lo, hi)])
        t0 = time.perf_counter()
        dfr = torch.from_numpy(fr).to(dev)
        torch.cuda.synchronize()
        h2d = (time.perf_counter() - t0) * 1e3
        ft = preprocess_videos_gpu([dfr], merge_size=2, max_tokens=16384 * (hi - lo) // budget_frames)
        _, gh_, gw_ = (int(v) for v in ft["grid_sizes"][0])
        return {"frames": fr, "dframes": dfr, "pix": ft["pixel_values"], "gh": gh_, "gw": gw_, "t_loc": hi - lo,
                "grid_loc": torch.tensor([[hi - lo, gh_, gw_]]), "h2d_ms": h2d, "T": T_clip}

    main_clip = make_clip(T, 256 if cfg3 else 64)
    frames, dframes, h2d_ms = main_clip["frames"], main_clip["dframes"], main_clip["h2d_ms"]
    pix = main_clip["pix"]        # bf16 on the GPU (evaluate/answer_generate.py:70 casts pixel_values to bf16)
    gh, gw, t_loc, grid_loc = main_clip["gh"], main_clip["gw"], main_clip["t_loc"], main_clip["grid_loc"]
    per_frame = gh * gw
    P = per_frame // 4
    n_patches, m_tokens = T * per_frame, T * P
    merge = torch.tensor([2])
    pix_all = None
    if world == 1:
        t0 = time.perf_counter()
        host = processing.preprocess_videos([frames], merge_size=2, max_tokens=16384 * t_loc // (256 if cfg3 else 64))
        t_host_pre = time.perf_counter() - t0
        pix_all = torch.from_numpy(host["pixel_values"])
        same_as_host = bool(torch.equal(pix.cpu(), pix_all.bfloat16()))

    # ---- weights (random, real dimensions) ----
    vit_state = random_vit_state(vcfg, seed=0, device=dev, dtype=torch.bfloat16)
    enc = VisionEncoder(vit_state, vcfg, dtype=torch.bfloat16, device=dev, fold_ln=False if args.no_ln_fold else None)
    proj = Projector(random_proj_state(vcfg.hidden_size, lcfg.hidden_size, seed=1, device=dev, dtype=torch.bfloat16),
                     dtype=torch.bfloat16, device=dev)
    del vit_state
    L.check(L.lib.cogs_vit_set_streams(enc.handle.h, args.vit_streams))
    wide = args.payload == "projected"

    def encode_step(clip, grid_full):
        """one step on this rank's frames of `clip` + the all-gather: projected tokens [M, 3584] on every rank"""
        if wide or world == 1:      # encoder + projector as the product path runs them (chat._encode_project): one library call
            return gather_tokens(enc.encode_project(clip["pix"], clip["grid_loc"], merge, proj)[1], grid_full, 2, world)
        tok = enc(clip["pix"], clip["grid_loc"], merge)
        return proj(gather_tokens(tok, grid_full, 2, world))

    def step():
        return encode_step(main_clip, (T, gh, gw))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(fn, warmup, steps):
        """W untimed + exactly K timed steps between barrier + synchronise on both sides; MAX over the ranks"""
        for _ in range(warmup):
            fn()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            r = fn()
        barrier()
        dt_ = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dt_], device="cpu" if rehearsal else dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt_ = float(tt)
        return dt_, r

    setup_s = time.perf_counter() - t_process0
    dt, mm = timed(step, args.warmup, args.steps)
    ms_per_step = dt / args.steps * 1e3
    fps = T * args.steps / dt
    gather_note = (f"one RCCL all-gather of the [M,{lcfg.hidden_size}] projected tokens" if wide else
                   f"one RCCL all-gather of the [M,{vcfg.hidden_size}] encoder tokens, projector on every rank")

    out = {
        "metric": "frames/sec encoded + answer tokens/sec, 64-frame clip, VideoLLaMA3-7B",
        "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
        "scaling": "weak" if weak else "strong", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic", "ranks_seen": ranks_seen,
        **({"debug_switches": debug_applied} if debug_applied else {}),
        "config": {"workload": f"{'cfg3 (BASELINE configs[2])' if cfg3 else ('cfg2 (BASELINE configs[1])' if T == 64 else 'cfg2-sized frames')}: "
                               f"{T}x480x854 '{args.clip}' clip -> {gh * 14}x{gw * 14}, "
                               f"{n_patches} patches, {m_tokens} visual tokens; ViT(1152x27, hd72)+projector(3584); "
                               f"random-init weights"
                               + (f"; WEAK scaling over {world} GPUs (the clip grows with N at {t_loc} frames per GPU)" if weak else
                                  (f"; STRONG scaling over {world} GPUs: the same fixed clip as at N = 1, {t_loc} frames per GPU" if world > 1 else "")),
                   "scaling_note": ("WEAK scaling: the clip grows with N" if weak else
                                    "STRONG scaling: the workload is the same fixed clip at every N (the N = 1 line is this curve's "
                                    "first point); side keys: 'cfg3' = BASELINE configs[2]'s 256-frame clip sharded the same way, "
                                    "'weak' = 64 frames per GPU"),
                   "frames": T, "frames_per_gpu": t_loc, "patches": n_patches, "visual_tokens": m_tokens,
                   "parallelism": (f"frames sharded over {world} GPUs ({t_loc} each), encode+project per rank, {gather_note}")
                   if world > 1 else "single GPU"},
    }

    # ---- N > 1: the all-gather on its own, both payloads (events on the stream the collective is queued from + wall) ----
    if world > 1:
        from cogstream_amd.parallel import gather_rows
        counts = [(e - b) * P for b, e in frame_shards(T, world)]
        ag = {"backend": dist.get_backend(), "note": "the one data-path collective of a step, alone: all-gather of this clip's "
              "visual tokens in frame order (parallel.gather_rows), MAX over ranks of the per-rank median of 10 calls, each "
              "between two barriers; ms_events = HIP events around the call on the current stream (the RCCL kernel is "
              "joined to it), ms_wall = host clock incl. launch" + ("; REHEARSAL on one GPU over gloo, staged through host "
              "memory: not a link measurement" if rehearsal else "")}
        for label, width in (("projected", lcfg.hidden_size), ("encoder", vcfg.hidden_size)):
            loc = torch.randn(counts[rank], width, device=dev, dtype=torch.float32).to(torch.bfloat16)
            for _ in range(3):
                gather_rows(loc, counts)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            tev, twall = [], []
            for _ in range(10):
                barrier()
                t0 = time.perf_counter()
                e0.record()
                gather_rows(loc, counts)
                e1.record()
                torch.cuda.synchronize()
                twall.append((time.perf_counter() - t0) * 1e3)
                tev.append(e0.elapsed_time(e1))
            med = torch.tensor([sorted(tev)[5], sorted(twall)[5]], dtype=torch.float64, device="cpu" if rehearsal else dev)
            dist.all_reduce(med, op=dist.ReduceOp.MAX)
            ag[label] = {"row_bytes": width * 2, "bytes_per_rank": counts[rank] * width * 2, "bytes_total": sum(counts) * width * 2,
                         "ms_events": round(float(med[0]), 4), "ms_wall": round(float(med[1]), 4),
                         "share_of_step": round(float(med[0]) / ms_per_step, 4)}
            del loc
        if rank == 0:
            out["allgather"] = ag
            out["allgather"]["payload_of_this_run"] = args.payload

    # ---- roofline of the dominant kernel (bf16 MFMA GEMM), HIP events around every launch ----
    # every rank runs this pass (it contains the all-gather); rank 0 reports its own kernels
    h = enc.handle
    ms = (C.c_float * 4)()
    cnt = (C.c_int * 4)()
    barrier()
    L.check(L.lib.cogs_profile_begin(h.h))
    step()
    L.check(L.lib.cogs_profile_end(h.h, L.current_stream(), ms, cnt))
    if rank == 0:
        n_loc, m_proj = t_loc * per_frame, t_loc * P
        gflops = vit_gemm_flops(n_loc, m_proj, vcfg, lcfg.hidden_size)
        gemm_ms, gemm_n = float(ms[0]), int(cnt[0])
        ach = gflops / (gemm_ms * 1e-3) / 1e12
        # HBM-side traffic per GEMM launch: NOT measured by this process (counters need their own rocprofv3 --pmc
        # passes: FETCH_SIZE / WRITE_SIZE in separate runs, gfx950 x2 read correction). The figure is read from the
        # PMC summary of the same command committed under profiles/ (tools/collect_profiles.sh), and labelled so.
        traffic, traffic_src = None, None
        for name in ("r6_gemm_traffic.json", "r5_gemm_traffic.json", "r4_gemm_traffic.json", "r3_gemm_traffic.json", "r2_gemm_traffic.json", "r1_g_gemm_traffic.json"):
            tpath = os.path.join(ROOT, "profiles", name)
            if world == 1 and T == 64 and not cfg3 and os.path.exists(tpath):
                traffic = json.load(open(tpath)).get("gemm_hbm_bytes_per_launch")
                traffic_src = f"committed PMC profile profiles/{name} (separate rocprofv3 --pmc passes of this command), not this run"
                break
        out["roofline"] = {"bound": "mfma",
                           "kernel": "bf16 MFMA GEMM (gemm_tn_pp64_kernel<*> + gemm_tn_256x128_kernel<*>: every encoder+"
                                     "projector GEMM kernel of one step; a round-aligned split counts as two launches)",
                           "achieved": round(ach, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                           "frac": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_src,
                           "launches": gemm_n, "avg_launch_ms": round(gemm_ms / max(gemm_n, 1), 4),
                           "flop_per_launch": gflops / max(gemm_n, 1),
                           # what a bare v_mfma_f32_16x16x32_bf16 loop on RANDOM operands delivers on this chip at the clock it
                           # holds (2.09-2.20 GHz under that load; 2 480 on zeros at 2.39 GHz): profiles/r6_mfma_shape_dvfs.txt
                           "bare_mfma_loop_tflops_random_data": BARE_MFMA_TFLOPS,
                           "frac_of_bare_mfma_loop": round(ach / BARE_MFMA_TFLOPS, 4),
                           "padding": "none since round 6 (MFMA busy cycles of the N = 1152 / 3456 GEMMs = their algorithmic "
                                      "count: profiles/r6_b_pmc_mfma_busy.txt), so MFMA-busy IS useful utilisation"}
        attn_ms = float(ms[1])
        out["breakdown_ms"] = {"gemm": round(gemm_ms, 2), "attention": round(attn_ms, 2), "norm": round(float(ms[2]), 2),
                               "other": round(float(ms[3]), 2)}
        out["attention_tflops"] = round(vit_attn_flops(t_loc, per_frame, vcfg) / (attn_ms * 1e-3) / 1e12, 1)
        total_flops = vit_gemm_flops(n_patches, m_tokens, vcfg, lcfg.hidden_size) + vit_attn_flops(T, per_frame, vcfg)
        out["encoder_tflops"] = round(total_flops / (ms_per_step * 1e-3) / 1e12, 1)   # whole job, all GPUs

    # ---- strong-scaling evidence on ONE GPU (--emulate-shard R, default 8): one rank's share of the clip -- frames
    # [0, T/R) -- encoded + projected alone on this GPU. shard_efficiency = (t_clip / R) / t_shard is what the encoder
    # stage of an R-GPU strong-scaling run can reach before the all-gather (1.0 = per-GPU time shrinks linearly) ----
    def time_steps(fn, n):
        fn()
        barrier()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        barrier()
        return (time.perf_counter() - t0) / n

    def shard_probe(pix_full, t_full, gh_, gw_, ms_full, R):
        t_sh = t_full // R
        pf = gh_ * gw_
        px = pix_full[: t_sh * pf]
        g = torch.tensor([[t_sh, gh_, gw_]])
        dt_sh = time_steps(lambda: enc.encode_project(px, g, merge, proj)[1], max(5, min(4 * args.steps, 20)))
        return {"ranks_emulated": R, "frames": t_sh, "patches": t_sh * pf, "ms_per_step": round(dt_sh * 1e3, 3),
                "frames_per_s_one_rank": round(t_sh / dt_sh, 1),
                "shard_efficiency": round((ms_full * 1e-3 / R) / dt_sh, 4),
                "note": "encode+project of one rank's frames on one GPU; the all-gather is not in it"}

    R = args.emulate_shard
    if rank == 0 and world == 1 and R > 1 and T % R == 0:
        out["shard%d" % R] = shard_probe(pix, T, gh, gw, ms_per_step, R)

    # ---- side curves, every rank takes part, rank 0 reports ----
    def side_curve(T_clip, budget, label):
        c = make_clip(T_clip, budget)
        g = (T_clip, c["gh"], c["gw"])
        n = max(3, min(args.steps, 10))
        dt_, mm_ = timed(lambda: encode_step(c, g), 1, n)
        pf = c["gh"] * c["gw"]
        fl = vit_gemm_flops(T_clip * pf, T_clip * pf // 4, vcfg, lcfg.hidden_size) + vit_attn_flops(T_clip, pf, vcfg)
        rec = {"workload": f"{label}: {T_clip}x480x854 '{args.clip}' clip -> {c['gh'] * 14}x{c['gw'] * 14}, {T_clip * pf} patches, "
                           f"{T_clip * pf // 4} visual tokens, {c['t_loc']} frames per GPU" + (", one all-gather" if world > 1 else ""),
               "value": round(T_clip * n / dt_, 2), "unit": "frames/s", "ms_per_step": round(dt_ / n * 1e3, 3), "steps": n,
               "n_gpus": world, "frames": T_clip, "frames_per_gpu": c["t_loc"], "encoder_tflops": round(fl / (dt_ / n) / 1e12, 1)}
        return rec, c, mm_, dt_ / n

    # BASELINE configs[2]: one 256-frame clip at the 16384-token budget, frames sharded over the ranks (32 per GPU at N = 8)
    if not cfg3 and not args.no_cfg3 and args.frames == 64 and not weak:
        rec3, c3, mm3, t3 = side_curve(256, 256, "BASELINE configs[2]")
        rec3["scaling"] = "strong"
        if rank == 0:
            out["cfg3"] = rec3
            if world == 1 and R > 1 and 256 % R == 0:
                out["cfg3"]["shard%d" % R] = shard_probe(c3["pix"], 256, c3["gh"], c3["gw"], t3 * 1e3, R)
        if rank == 0 and world == 1:
            out["stages"] = stage_rooflines(c3, mm3, dev, with_cpu=not args.no_cpu)
        del c3, mm3
    # the weak curve: 64 frames per GPU (N = 4 is a 256-frame clip at cfg2's frame size)
    if world > 1 and not cfg3 and not weak and args.frames == 64:
        recw, cw, mmw, _ = side_curve(64 * world, 64, "cfg2-sized frames, clip grows with N")
        recw["scaling"] = "weak"
        if rank == 0:
            out["weak"] = recw
        del cw, mmw

    # ---- Qwen2-7B: prefill of the interleaved prompt + greedy decode (rank 0; other ranks wait) ----
    if rank == 0 and world == 1 and not args.no_llm:
        from cogstream_amd.llm import Qwen2Engine
        enc.handle._ws.pop("vit", None)  # release the encoder workspace
        torch.cuda.empty_cache()
        eng = Qwen2Engine(random_llm_state(lcfg, seed=2, device=dev, dtype=torch.bfloat16), lcfg,
                          dtype=torch.bfloat16, device=dev)
        torch.cuda.empty_cache()
        # prompt layout of the real tokenizer (prompt_tokens above); text rows are random embeddings
        S = prompt_tokens(T, P)
        embeds = torch.randn(S, lcfg.hidden_size, device=dev, dtype=torch.float32).mul_(0.02).to(torch.bfloat16)
        pos = 33
        for f in range(T):
            hdr = frame_header_tokens(f)
            embeds[pos + hdr:pos + hdr + P] = mm[f * P:(f + 1) * P]
            pos += hdr + P + 1
        ndec = args.decode_tokens
        # warm-up at FULL size (allocations, first touch of the 15k-token workspace, kernel code objects): the timed
        # prefill and the timed token loop below both run warm, and each is timed on its own
        cache = eng.new_cache(S + ndec + 8)
        eng.generate(embeds, max_new_tokens=4, ignore_eos=True, cache=cache)
        torch.cuda.synchronize()
        cache.reset(0)
        t0 = time.perf_counter()
        res = eng.forward(embeds, cache)
        torch.cuda.synchronize()
        t_prefill = time.perf_counter() - t0
        t0 = time.perf_counter()
        toks = eng.generate(embeds, max_new_tokens=ndec, repetition_penalty=1.05, ignore_eos=True, cache=cache,
                            prefilled=res)
        torch.cuda.synchronize()
        t_dec = time.perf_counter() - t0          # the token loop alone: ndec - 1 forwards + ndec token selections
        n_fwd = len(toks) - 1
        bytes_per_token = 2 * (6.526e9 + 545e6) + 57344.0 * (S + n_fwd / 2)      # weights + K/V rows
        tbps = n_fwd * bytes_per_token / t_dec / 1e12
        out["answer_tokens_per_s"] = round(n_fwd / t_dec, 2)
        out["llm"] = {"prompt_tokens": S, "prefill_s": round(t_prefill, 4), "decode_tokens": len(toks),
                      "decode_s": round(t_dec, 4), "ms_per_token": round(t_dec / n_fwd * 1e3, 4), "prefill_tflops": round(
                          (2.0 * S * 6.526e9 + 2.0 * S * S * lcfg.hidden_size * lcfg.num_hidden_layers / 2) / t_prefill / 1e12, 1),
                      "timing": "prefill and token loop timed separately, both after a full-size warm-up"}
        dtraffic, dsrc = None, None
        dname = next((n for n in ("r6_decode_traffic.json", "r3_decode_traffic.json")
                      if os.path.exists(os.path.join(ROOT, "profiles", n))), None)
        if dname:                       # fabric-side read bytes of one decode step, from a separate rocprofv3 --pmc pass
            dtraffic = json.load(open(os.path.join(ROOT, "profiles", dname))).get("fetch_bytes_per_token")
            dsrc = (f"committed PMC profile profiles/{dname} (rocprofv3 --pmc FETCH_SIZE on tools/decode_trace.py "
                    "at the same context, tools/decode_traffic.py; FETCH doubled per the gfx950 correction), not this run")
        out["decode"] = {"roofline": {"bound": "hbm", "bytes_per_token": int(bytes_per_token),
                                      "traffic": dtraffic, "traffic_source": dsrc,
                                      "achieved_TBps": round(tbps, 3), "peak_TBps": HBM_PEAK_TBPS,
                                      "frac_of_8TBps": round(tbps / HBM_PEAK_TBPS, 4),
                                      "kernel": "gemv_kernel<*> weight streaming + attn_decode_kernel (csrc/gemv.hip, "
                                                "csrc/attn_decode.hip); wall time of the token loop, launches included"}}
        out["e2e_s"] = round(ms_per_step * 1e-3 + t_prefill + t_dec, 4)
        # the reference's SHIPPED generation mode (model/generation_config.json:2-12: do_sample, temperature 0.7,
        # top_k 20, top_p 0.8, repetition_penalty 1.05), sampled on the device (cogs_sample, Philox draws)
        cache.reset(0)
        res = eng.forward(embeds, cache)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        toks_s = eng.generate(embeds, max_new_tokens=ndec, do_sample=True, temperature=0.7, top_k=20, top_p=0.8,
                              repetition_penalty=1.05, ignore_eos=True, cache=cache, seed=1234, prefilled=res)
        torch.cuda.synchronize()
        t_dec_s = time.perf_counter() - t0
        out["answer_tokens_per_s_sampled"] = round((len(toks_s) - 1) / t_dec_s, 2)
        out["llm"]["sampled"] = {"config": "do_sample T=0.7 top_k=20 top_p=0.8 repetition_penalty=1.05 (generation_config.json)",
                                 "decode_tokens": len(toks_s), "decode_s": round(t_dec_s, 4), "sampler": "device (cogs_sample, Philox)"}
        del cache
        torch.cuda.empty_cache()
        # ---- the whole product path, driver-timed: CogStreamProcessor -> qa_selection -> generate (64 new tokens,
        # greedy) with the real tokenizer's prompt lengths, at cfg2 (this clip) and cfg3 (256 frames: k-means +
        # 18 event-summary passes); evaluate/answer_generate.py:60-76 ----
        if not args.no_pipeline and T == 64 and not cfg3:
            from cogstream_amd.chat import CogReasoner
            model = CogReasoner(enc, proj, eng, lcfg, generation_config=dict(do_sample=False, eos_token_id=[]))
            processor = processing.CogStreamProcessor(PromptLengthTokenizer(), device=dev)
            pipeline_once(model, processor, dframes, 4)                       # warm-up
            out["pipeline"] = {"cfg2": pipeline_once(model, processor, dframes, 64, expect_prompt=prompt_tokens(64, P)),
                               "note": "wall time of one answer through the product API (evaluate/answer_generate.py:60-76), raw "
                                       "uint8 frames already on the GPU; stages_s: each stage drained on both sides; "
                                       "host_and_small_kernels = the rest (tokenisation, index bookkeeping, pixel-diff mask, "
                                       "compaction, splice)"}
            clip256 = torch.from_numpy(np.concatenate([processing.synthetic_clip(64, kind="drift", clip_idx=c)[0]
                                                       for c in range(4)])).to(dev)
            pipeline_once(model, processor, clip256, 4)
            out["pipeline"]["cfg3"] = pipeline_once(model, processor, clip256, 64, expect_prompt=prompt_tokens(256, 50))
            out["pipeline"]["cfg3"]["clip"] = "drift (the noise clip keeps every token; drift prunes by pixel difference)"
            del clip256, model
            if not args.no_session:
                out["session"] = session_probe(
                    lambda: CogReasoner(enc, proj, eng, lcfg, generation_config=dict(do_sample=False, eos_token_id=[])),
                    processor, dev)
        del eng
        torch.cuda.empty_cache()
        # ---- CPU column for the token rate: the oracle's Qwen2 (torch fp32) decoding ONE token at the same context on
        # the host cores -- 2 of the 28 layers timed and scaled, plus the lm_head (a bounded sample: ~10-20 s) ----
        if not args.no_cpu:
            from oracle import qwen2 as oq
            ncpu = min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
            torch.set_num_threads(ncpu)
            l2 = LlmConfig(num_hidden_layers=2)
            wc = random_llm_state(l2, seed=2, device="cpu", dtype=torch.float32)
            kvh, hd = l2.num_key_value_heads, l2.head_dim
            past = [(torch.randn(kvh, S, hd), torch.randn(kvh, S, hd)) for _ in range(2)]
            e1 = torch.randn(1, l2.hidden_size) * 0.02
            with torch.no_grad():
                oq.forward(wc, e1, heads=l2.num_attention_heads, kv_heads=kvh, layers=2, past=past)       # warm-up
                n_tok = 3
                t0 = time.perf_counter()
                for _ in range(n_tok):
                    hid, _ = oq.forward(wc, e1, heads=l2.num_attention_heads, kv_heads=kvh, layers=2, past=past)
                t_layers = (time.perf_counter() - t0) / n_tok
                t0 = time.perf_counter()
                for _ in range(n_tok):
                    oq.logits(wc, hid[-1])
                t_head = (time.perf_counter() - t0) / n_tok
            t_tok = t_layers * (lcfg.num_hidden_layers / 2) + t_head
            out["decode"]["cpu_baseline"] = {"value": round(1.0 / t_tok, 3), "unit": "tokens/s", "cores": ncpu, "kind": "port",
                                             "sample": f"oracle.qwen2.forward (torch fp32) decoding one token at context {S}: 2 of "
                                                       f"{lcfg.num_hidden_layers} layers timed ({t_layers * 1e3:.0f} ms) and scaled x"
                                                       f"{lcfg.num_hidden_layers // 2}, + lm_head ({t_head * 1e3:.0f} ms); {n_tok} tokens"}
            del wc, past

    # ---- GPU pre-processing of the clip (uint8 frames -> pixel_values) ----
    if rank == 0 and world == 1:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            preprocess_videos_gpu([dframes])
        torch.cuda.synchronize()
        out["preprocess"] = {"gpu_ms": round((time.perf_counter() - t0) / 5 * 1e3, 3), "h2d_ms": round(h2d_ms, 3),
                             "host_pil_ms": round(t_host_pre * 1e3, 1), "equal_to_host_path": same_as_host}

    # ---- CPU baseline: the oracle on host cores, bounded sample (rank 0, N = 1) ----
    if rank == 0 and world == 1 and not args.no_cpu:
        from oracle import vision as ov
        from cogstream_amd.weights import random_vit_state as rvs
        # the GPU box gives one GPU a 16-core CPU share; more threads than that only thrash
        ncpu = min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
        torch.set_num_threads(ncpu)
        st = rvs(vcfg, seed=0, device="cpu", dtype=torch.float32)
        nfr = 8   # ~10-15 s of CPU work on 16 cores (bounded sample)
        px = pix_all[:nfr * per_frame].float()
        g = torch.tensor([[nfr, gh, gw]])
        with torch.no_grad():
            ov.encode(st, px[:per_frame], torch.tensor([[1, gh, gw]]), merge, heads=vcfg.num_attention_heads,
                      layers=2)  # warm-up of the thread pool
            t0 = time.perf_counter()
            ov.encode(st, px, g, merge, heads=vcfg.num_attention_heads, layers=vcfg.num_hidden_layers)
            tc = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": round(nfr / tc, 3), "unit": "frames/s", "cores": ncpu, "kind": "port",
                               "sample": f"oracle.vision.encode (torch fp32, block-diagonal) on {nfr} frames of the same "
                                         f"clip ({nfr * per_frame} patches, 27 layers), {tc:.1f} s"}

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        out["wall_s"] = {"setup_before_first_step": round(setup_s, 2), "rank_process_total": round(time.perf_counter() - t_process0, 2),
                         "note": "seconds inside this rank process (interpreter start and `import torch` come before it)"}
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
