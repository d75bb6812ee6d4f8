"""CogReasoner: host mirror of Videollama3Qwen2ForCausalLM's inference surface
(model/cogreasoner_chat.py), same method names / argument meaning / return shapes, computing through the
HIP C ABI:

    qa_selection(...)                         :809-865   Historic Dialogue Retrieval + prompt surgery
    generate(...) -> (ids [1,n], selection)   :753-807
    prepare_inputs_labels_for_multimodal      :513-584   encode -> cluster -> compress -> embed
    encode_images / select_events_based_on_summary / compress_unimportant_events /
    _get_compression_mask / _compress_visual_tokens / prepare_inputs / generate_language_module

Only control flow, string handling and index bookkeeping live here; every tensor op is a kernel
(cogstream_amd.ops / vision / llm / kmeans). The object is stateful between qa_selection() and generate()
exactly like the reference (:812-816)."""
from __future__ import annotations

import math
import re
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import ops
from .kmeans import kmeans_with_time_min_max, select_additional_frames
from .llm import Qwen2Engine
from .qaselect import select_qas
from .vision import Projector, VisionEncoder
from .weights import LlmConfig

SUMMARY_INSTRUCTION = (
    "Concisely list the key points of the event shown in the timestamped images, adhering strictly and honestly to "
    "the visual content. For each key point, identify relevant objects or actions, note any visible text, and specify "
    "the approximate timestamp(s). Provide an overview focusing on these key timestamped points.")
SUMMARY_SYSTEM = "You are a helpful assistant specializing in summarizing events from timestamped visual data."

# model/generation_config.json:2-12
DEFAULT_GENERATION = dict(do_sample=True, temperature=0.7, top_k=20, top_p=0.8, repetition_penalty=1.05,
                          eos_token_id=[151645, 151643], pad_token_id=151643)

FRAMES_PER_EVENT = 15      # :280
MIN_EVENTS = 9             # :281  (k-means only when ceil(T/15) > 9)
EXTRA_FRAMES = 2           # :284
COSINE_THRESHOLD = 0.45    # :329


def create_visual_summary_prompt(P: int, timestamps, image_token: str = "<image>") -> str:
    """:93-119 -- P = total visual tokens of the event, one 'Time x.xs:' + P//T placeholders per frame"""
    T = len(timestamps)
    per = image_token * (P // T)
    frames = []
    for t in range(T):
        v = timestamps[t].item() if isinstance(timestamps[t], torch.Tensor) else float(timestamps[t])
        frames.append(f"Time {v:.1f}s:{per}" + ("," if t < T - 1 else ""))
    return (f"<|im_start|>system\n{SUMMARY_SYSTEM}<|im_end|>\n<|im_start|>user\n" + "".join(frames) + "\n"
            + SUMMARY_INSTRUCTION + "<|im_end|>\n<|im_start|>assistant")


_TIME_RUN_COMMA = re.compile(r"Time \d+\.\d+s:(?:<image>)*,")
_TIME_RUN_NL = re.compile(r"Time \d+\.\d+s:(?:<image>)*\n")
_VISUAL_PREFIX = re.compile(r"((?:(?:Time \d+\.\d+s:(?:<image>)*),?)*)\s*(.*)")


def process_input_ids(text: str, if_visual: bool, hist_qs: Sequence[str], hist_as: Sequence[str],
                      current_question: str, tokenizer=None) -> str:
    """:121-177 -- keep the system turn, the current question, and only the selected history turns; when
    the selector said 'no', strip every 'Time x.xs:<image>...' run first; a dropped question's visual
    content is kept and glued to the next kept turn (appendix A quirk)."""
    if not if_visual:
        text = _TIME_RUN_COMMA.sub("", text)
        text = _TIME_RUN_NL.sub("", text)
    kept: List[str] = []
    for seg in text.split("<|im_start|>")[1:]:
        rc = seg.split("\n", 1)
        if len(rc) != 2:
            continue
        role, content = rc[0].strip(), rc[1].split("<|im_end|>")[0].strip()
        full = f"<|im_start|>{role}\n{content}<|im_end|>\n"
        if role == "system":
            kept.append(full)
        elif role == "user":
            visual, question = "", content
            if if_visual:
                m = _VISUAL_PREFIX.match(content)
                if m:
                    visual, question = m.group(1).rstrip(",").strip(), m.group(2).strip()
            if question == current_question or question in hist_qs:
                kept.append(full)
            elif if_visual and visual:
                kept.append(f"<|im_start|>{role}\n{visual}")
        elif role == "assistant":
            if content in hist_as:
                kept.append(full)
    kept.append("<|im_start|>assistant\n")
    out: List[str] = []
    for i, seg in enumerate(kept):
        head = "<|im_start|>user\n"
        if seg.startswith(head) and (i == 0 or not kept[i - 1].rstrip().endswith("<|im_end|>")):
            body = seg[len(head):]
            if body.strip():
                out.append(body)
            continue
        out.append(seg)
    return "".join(out)


def parse_selection(selection_module_output: str):
    """:479-499 -> (if_visual, [indices])"""
    if_visual = True
    parts = selection_module_output.strip("[]").split(",")
    if parts and parts[0]:
        first = parts[0].strip()
        if first == "no":
            if_visual = False
            parts = parts[1:]
        elif first == "yes":
            parts = parts[1:]
    idx = []
    for p in parts:
        p = p.strip()
        if p:
            try:
                idx.append(int(p))
            except ValueError:
                continue
    return if_visual, idx


class _Stage:
    def __init__(self, owner, name):
        self.times, self.name = getattr(owner, "stage_times", None), name

    def __enter__(self):
        if self.times is not None:
            import time
            torch.cuda.synchronize()
            self.t0 = time.perf_counter()

    def __exit__(self, *exc):
        if self.times is not None:
            import time
            torch.cuda.synchronize()
            self.times[self.name] = self.times.get(self.name, 0.0) + time.perf_counter() - self.t0
        return False


class CogReasoner:
    def __init__(self, vision: VisionEncoder, projector: Projector, llm: Qwen2Engine, config: Optional[LlmConfig] = None,
                 generation_config: Optional[dict] = None, use_token_compression: bool = True):
        self.vision_encoder, self.mm_projector, self.llm = vision, projector, llm
        if llm is None and config is None:
            raise ValueError("an encoder-only helper (llm=None, see enable_sharded_encoder) still needs the LlmConfig")
        self.config = config or llm.cfg
        owner = llm if llm is not None else vision
        self.device, self.dtype = owner.device, owner.dtype
        self._shard = None
        self.stage_times: Optional[Dict[str, float]] = None     # set to {} to collect a per-stage wall-time split
        self.generation_config = dict(DEFAULT_GENERATION if generation_config is None else generation_config)
        self.use_token_compression = use_token_compression
        self.tokenizer = None
        self.hist_qs: List[str] = []
        self.hist_as: List[str] = []
        self.current_question = ""
        self.all_timestamps: Optional[torch.Tensor] = None
        self.last_debug: Dict[str, object] = {}
        self.cosine_override: Optional[Sequence[float]] = None
        self._adapters: Dict[str, tuple] = {"base": (llm, projector)}
        self.active_adapter = "base"

    # ------------------------------------------------------------------ loading
    @classmethod
    def from_pretrained(cls, path: str, torch_dtype=None, device=None, attn_implementation: str = "flash_attention_2",
                        trust_remote_code: bool = True, load_llm: bool = True, **unused) -> "CogReasoner":
        """AutoModelForCausalLM.from_pretrained(model_path, trust_remote_code=True, torch_dtype=torch.bfloat16,
        attn_implementation="flash_attention_2") of evaluate/answer_generate.py:173-178 for this implementation:
        config.json / generation_config.json / model.safetensors.index.json + shards of the checkpoint directory.
        attn_implementation: "flash_attention_2" = per-frame (block-diagonal) ViT attention, the path the reference
        ships with on GPUs; "eager" = the reference's CPU semantics (global attention + same-frame bias,
        modeling_videollama3_encoder.py:257-266), for parity runs; "sdpa" is broken in the reference (:348) and
        refused here. The model is built straight on `device` (default cuda:<current>): the later .to(local_rank)
        of the reference driver (:183) is then a no-op."""
        from . import checkpoint as ck
        from .vision import BLOCK_DIAG, REF_EAGER_GLOBAL
        modes = {"flash_attention_2": BLOCK_DIAG, "eager": REF_EAGER_GLOBAL}
        if attn_implementation not in modes:
            raise ValueError(f"attn_implementation={attn_implementation!r}: use 'flash_attention_2' (block-diagonal) or 'eager'")
        cfgs = ck.load_configs(path)
        if torch_dtype is None:
            torch_dtype = getattr(torch, cfgs["torch_dtype"], torch.bfloat16)
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if dev.type == "cuda" and dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        reader = ck.Checkpoint(path, device=dev)
        want = ck.expected_tensors(cfgs["vision"], cfgs["llm"], cfgs["tie_word_embeddings"])
        lacking = sorted(set(want) - set(reader.weight_map))
        if lacking:
            raise RuntimeError(f"{path}: checkpoint lacks {len(lacking)} tensors of the module tree, e.g. {lacking[:4]}")
        reader.check_shapes(want)
        vit_v, proj_v, llm_v = ck.state_views(reader, cfgs["tie_word_embeddings"])
        enc = VisionEncoder(vit_v, cfgs["vision"], dtype=torch_dtype, device=dev, attn_mode=modes[attn_implementation])
        proj = Projector(proj_v, dtype=torch_dtype, device=dev)
        # load_llm=False: an encoder-only helper rank of a frame-sharded run (enable_sharded_encoder): the 15 GB of
        # Qwen2 weights are neither read nor held; everything else of the checkpoint must still be consumed
        eng = Qwen2Engine(llm_v, cfgs["llm"], dtype=torch_dtype, device=dev) if load_llm else None
        reader.check_consumed(extra_ok=() if load_llm else tuple(llm_v._full(k) for k in llm_v))
        reader.close()
        # HF: generation_config.json, when present, IS the generation config (nothing is merged in from elsewhere);
        # without the file generate() runs on GenerationConfig defaults (greedy, no penalty)
        gen = dict(cfgs["generation"]) if cfgs["generation"] else dict(do_sample=False, eos_token_id=[cfgs["llm"].eos_token_id])
        model = cls(enc, proj, eng, cfgs["llm"], generation_config=gen, use_token_compression=cfgs["use_token_compression"])
        model.name_or_path = path
        return model

    def to(self, device=None, *a, **k) -> "CogReasoner":
        """nn.Module.to of the reference driver (answer_generate.py:183): weights already live on the device chosen at
        load time; moving a packed model between GPUs is not supported"""
        if device is not None and not isinstance(device, torch.dtype):
            d = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
            if d.type != "cuda" or (d.index is not None and d.index != self.device.index):
                raise ValueError(f"model was loaded on {self.device}; load it with from_pretrained(device={d})")
        return self

    def eval(self) -> "CogReasoner":
        return self

    def load_adapter_from_path(self, path: str, adapter_name: str) -> None:
        """PeftModel.from_pretrained(model, path, adapter_name=...) / model.load_adapter(path, adapter_name=...)
        (answer_generate.py:181-182): peft adapter directory -> one merged weight set (weights.merge_lora)"""
        from . import checkpoint as ck
        base = getattr(self, "name_or_path", None)
        if base is None:
            raise ValueError("load_adapter_from_path needs a model made by from_pretrained (the base weights are re-read)")
        lora, alpha = ck.load_adapter_state(path, device=self.device)
        reader = ck.Checkpoint(base, device=self.device)
        _, proj_v, llm_v = ck.state_views(reader)
        self.load_adapter(dict(llm_v), dict(proj_v), lora, adapter_name, lora_alpha=alpha)
        reader.close()

    # ------------------------------------------------------------------ adapters
    # The peft surface evaluate/answer_generate.py uses (:71-73 set_adapter, :181-182 from_pretrained /
    # load_adapter): an adapter here is a merged copy of the Qwen2 (and projector) weights -- weights.merge_lora --
    # so switching re-points the engine and costs no arithmetic.
    def add_adapter(self, adapter_name: str, llm: Qwen2Engine, projector: Optional[Projector] = None) -> None:
        self._adapters[adapter_name] = (llm, projector if projector is not None else self._adapters["base"][1])

    def load_adapter(self, llm_state, proj_state, lora_state, adapter_name: str, lora_alpha: float = 16.0) -> None:
        """llm_state / proj_state: the BASE weights in HF layout (tensors already on the device are shared, not
        copied, where the adapter does not touch them); lora_state: peft adapter_model state dict"""
        from .weights import merge_lora
        ml, mp = merge_lora(llm_state, proj_state, lora_state, self.config, lora_alpha)
        llm = Qwen2Engine(ml, self.config, dtype=self.dtype, device=self.device)
        proj = Projector(mp, dtype=self.dtype, device=self.device) if mp is not None else None
        self.add_adapter(adapter_name, llm, proj)

    def set_adapter(self, adapter_name: str) -> None:
        if adapter_name not in self._adapters:
            raise ValueError(f"Adapter {adapter_name} not found.")   # peft's wording
        self.llm, self.mm_projector = self._adapters[adapter_name]
        self.active_adapter = adapter_name

    # ------------------------------------------------------------------ vision
    def enable_distributed_events(self, rank: int, world: int, group=None) -> None:
        """spread the K + 1 event-summary sequences of select_events_based_on_summary over `world` ranks that all hold
        this model and the same inputs (same host RNG state for the k-means seeding): parallel.pooled_means_sharded"""
        self._event_rank, self._event_world, self._event_group = int(rank), int(world), group

    def enable_sharded_encoder(self, rank: int, world: int, group=None, payload: str = "projected",
                               llm_rank: int = 0) -> None:
        """BASELINE configs[2]: the frames of a request shard data-parallel over `world`
        processes (one per GPU) that all call qa_selection() / generate() with the SAME inputs -- the reference's
        launcher contract, evaluate/answer_generate.py:154-158,169-183 (one process per GPU, LOCAL_RANK). Inside
        encode_images (:264-276) rank r encodes only its contiguous run of frames (parallel.FramePlan) and ONE
        all-gather (RCCL over xGMI; gloo in rehearsals) reassembles the visual tokens in frame order on every rank.
          payload "projected": project the local tokens, gather [M, 3584]  (projector work done once, 3.1x the bytes)
          payload "encoder":   gather the encoder's [M, 1152] tokens, every rank projects all of them
        The LLM stages run on `llm_rank` only: it owns the retrieval decode of qa_selection (the selection string is
        broadcast), k-means (assignments broadcast when enable_distributed_events() spreads the event-summary passes,
        else nobody else needs them), the pixel-difference mask on its full pixel_values, prefill and decode; the new
        token ids are broadcast so that every rank returns the same answer and multi-turn conversations stay in step.
        Ranks other than llm_rank may be built without the Qwen2 weights (llm=None / from_pretrained(load_llm=False))
        unless the event passes are spread. The reference's CPU attention semantics (attn_implementation="eager":
        every patch attends to every frame) do not shard: in that mode every rank encodes the whole request."""
        if payload not in ("projected", "encoder"):
            raise ValueError(f"payload={payload!r}: 'projected' (3584-wide gather) or 'encoder' (1152-wide gather)")
        if not (0 <= int(rank) < int(world) and 0 <= int(llm_rank) < int(world)):
            raise ValueError(f"rank {rank} / llm_rank {llm_rank} outside world {world}")
        if int(rank) == int(llm_rank) and self.llm is None:
            raise ValueError("the llm_rank needs the Qwen2 weights")
        self._shard = None if int(world) == 1 else dict(rank=int(rank), world=int(world), group=group, payload=payload,
                                                        llm_rank=int(llm_rank))

    def _stage(self, name: str):
        """`with self._stage("encode"):` -- when self.stage_times is a dict, the wall time of the block (device drained
        on both sides) is added to stage_times[name]; otherwise free. bench.py's `pipeline` split."""
        return _Stage(self, name)

    def _is_helper(self) -> bool:
        return self._shard is not None and self._shard["rank"] != self._shard["llm_rank"]

    def _bcast(self, obj, group=None, src: Optional[int] = None):
        """small python object from one rank (default: the llm_rank of the sharded run) to every rank of the group"""
        import torch.distributed as dist
        if src is None:
            group, src = self._shard["group"], self._shard["llm_rank"]
        box = [obj]
        dist.broadcast_object_list(box, src=src if group is None else dist.get_global_rank(group, src), group=group)
        return box[0]

    def _encode_project(self, pixel_values, grid_sizes, merge_sizes) -> torch.Tensor:
        """vision_encoder + mm_projector of :270-275 -- on this rank's frames + one all-gather when sharded"""
        from .vision import REF_EAGER_GLOBAL
        sh = self._shard
        if sh is None or self.vision_encoder.attn_mode == REF_EAGER_GLOBAL:
            return self.vision_encoder.encode_project(pixel_values, grid_sizes, merge_sizes, self.mm_projector)[1]
        from .parallel import FramePlan, gather_rows
        plan = FramePlan(grid_sizes, merge_sizes, sh["world"])
        px, g, m = plan.local(pixel_values, sh["rank"])
        wide = sh["payload"] == "projected"
        width = self.mm_projector.packed.out_dim if wide else self.vision_encoder.cfg.hidden_size
        if g.shape[0]:
            if wide:
                tok = self.vision_encoder.encode_project(px, g, m, self.mm_projector)[1]
            else:
                tok = self.vision_encoder(px, g, m)
        else:                                   # more ranks than frames: nothing to encode, still in the collective
            tok = torch.empty(0, width, device=self.device, dtype=self.dtype)
        self.last_debug.update(shard_pieces=plan.pieces[sh["rank"]], shard_tokens=plan.token_counts)
        out = gather_rows(tok, plan.token_counts, sh["group"])
        return out if wide else self.mm_projector(out)

    def enable_prefix_cache(self, on: bool = True) -> None:
        """keep the KV rows of the previous answer prompt and of the previous retrieval prompt (one slot per stage
        and adapter) and prefill only the rows that changed (llm.PrefixKV)"""
        self._prefix_on = bool(on)
        self._prefix_slots = {}

    def _prefix_slot(self, stage: str):
        if not getattr(self, "_prefix_on", False):
            return None
        from .llm import PrefixKV
        key = (stage, id(self.llm))   # set_adapter() swaps the engine: KV rows of one weight set are not another's
        slot = self._prefix_slots.get(key)
        if slot is None:
            slot = self._prefix_slots[key] = PrefixKV(self.llm)
        return slot

    def prefix_cache_stats(self):
        """{stage: (rows reused, prompt rows seen)} summed over the weight sets"""
        out = {}
        for (stage, _), v in getattr(self, "_prefix_slots", {}).items():
            r, n = out.get(stage, (0, 0))
            out[stage] = (r + v.reused, n + v.seen)
        return out

    def enable_visual_cache(self, max_videos: int = 64) -> None:
        """Streaming-session cache: projected visual tokens per video segment, keyed
        by (content key from the processor, grid t x h x w, merge size, projector in use). The reference re-encodes
        every segment on every turn (evaluate/answer_generate.py:130-148 rebuilds the whole conversation); frames
        are independent under block-diagonal attention and the projector is per token, so a cached segment is
        bit-identical to a recomputed one. A segment whose resize target drifted (the token budget is shared by
        all frames of the conversation) has a different grid and simply misses."""
        from collections import OrderedDict
        self._vcache = OrderedDict()
        self._vcache_max = max_videos
        self.visual_cache_stats = {"hits": 0, "misses": 0}

    def encode_images(self, pixel_values, grid_sizes, merge_sizes, video_keys=None) -> torch.Tensor:
        """:264-276"""
        cache = getattr(self, "_vcache", None)
        if cache is None or video_keys is None or len(video_keys) != int(grid_sizes.shape[0]):
            return self._encode_project(pixel_values, grid_sizes, merge_sizes)
        gs = [tuple(int(x) for x in g) for g in grid_sizes.tolist()]
        ms = [int(m) for m in merge_sizes.tolist()]
        rows = [t * h * w for t, h, w in gs]
        offs = [0]
        for r in rows:
            offs.append(offs[-1] + r)
        keys = [(k, g, m, id(self.mm_projector)) for k, g, m in zip(video_keys, gs, ms)]
        miss = [v for v, k in enumerate(keys) if k not in cache]
        self.visual_cache_stats["hits"] += len(keys) - len(miss)
        self.visual_cache_stats["misses"] += len(miss)
        if miss:
            px = pixel_values if len(miss) == len(keys) else torch.cat([pixel_values[offs[v]:offs[v + 1]] for v in miss])
            tok = self._encode_project(px, grid_sizes[miss], merge_sizes[miss])
            o = 0
            for v in miss:
                n = rows[v] // (ms[v] * ms[v])
                cache[keys[v]] = tok[o:o + n].clone()
                o += n
        for k in keys:
            cache.move_to_end(k)
        out = torch.cat([cache[k] for k in keys], dim=0) if len(keys) > 1 else cache[keys[0]].clone()
        while len(cache) > self._vcache_max:   # callers edit the result in place (event compression): always a copy
            cache.popitem(last=False)
        return out

    def _bf16_round(self, x: torch.Tensor) -> torch.Tensor:
        return x.to(torch.bfloat16).to(torch.float32) if self.dtype == torch.bfloat16 else x

    def _pooled_forward(self, idx: torch.Tensor, mm_features: Optional[torch.Tensor]) -> torch.Tensor:
        emb = ops.gather_rows(self.llm.packed.embed, mm_features, idx.to(self.device))
        pooled = self.llm.forward(emb, None, want_logits=False, want_pooled=True)["pooled"]
        return self._bf16_round(pooled)  # torch.mean of a bf16 tensor is bf16

    def select_events_based_on_summary(self, mm_features: torch.Tensor, total_image_num: int, timestamps) -> List[int]:
        """:278-333 -- returns the (ascending) frame indices of events that are irrelevant to the question,
        minus the 2 frames nearest each event centroid"""
        T = total_image_num
        P, D = mm_features.shape[0] // T, mm_features.shape[1]
        K = math.ceil(T / FRAMES_PER_EVENT)
        if K <= MIN_EVENTS:
            return []
        spread = getattr(self, "_event_world", 1) > 1
        if self._is_helper() and not spread:
            return []        # frame-sharded run: k-means, the event passes and the decision belong to the llm_rank
        features = mm_features.view(T, P, D)
        # spread event passes: ONE rank clusters (the host RNG draws of k-means are that rank's, as in a one-process
        # run) and broadcasts the T assignments + the near-centroid picks; every rank then builds the same sequences
        root = self._shard["llm_rank"] if self._shard is not None else 0
        if not spread or self._event_rank == root:
            with self._stage("kmeans"):
                centres, _, assign = kmeans_with_time_min_max(features, timestamps, K)
                picked = select_additional_frames(features, centres, assign, EXTRA_FRAMES)
                picked_set = set(torch.cat(picked, dim=0).view(-1).tolist())
                assign_h = assign.cpu().tolist()
        if spread:
            if self.llm is None:
                raise RuntimeError("enable_distributed_events() needs the Qwen2 weights on every rank")
            assign_h, picked_set = self._bcast((assign_h, picked_set) if self._event_rank == root else None,
                                               self._event_group, root)
        ts = timestamps.cpu() if isinstance(timestamps, torch.Tensor) else torch.tensor(timestamps)
        image_id = self.config.image_token_index
        # the K event prompts and the question are K+1 independent sequences: the reference runs one forward per
        # sequence (:303-322), here they are ONE var-len prefill with per-sequence mean pooling
        segs = []
        for k in range(K):
            frames = [i for i in range(T) if assign_h[i] == k]
            prompt = create_visual_summary_prompt(len(frames) * P, [ts[i] for i in frames])
            ids = self.tokenizer(prompt, return_tensors="pt")["input_ids"].reshape(-1).to(torch.int64)
            rows = (torch.tensor(frames, dtype=torch.int64)[:, None] * P + torch.arange(P, dtype=torch.int64)[None, :]).reshape(-1)
            sel = ids == image_id
            assert int(sel.sum()) == rows.numel()
            idx = ids.clone()
            idx[sel] = -(rows + 1)
            segs.append(idx)
        q = self.tokenizer(self.current_question, padding=True, truncation=True, return_tensors="pt", max_length=128)
        segs.append(q["input_ids"].reshape(-1).to(torch.int64))
        lens = [int(x.numel()) for x in segs]
        self.last_debug.update(event_tokens=sum(lens))
        with self._stage("event_prefill"):
            emb = ops.gather_rows(self.llm.packed.embed, mm_features, torch.cat(segs).to(self.device))
            if spread:   # enable_distributed_events(): the sequences are spread over the ranks
                from .parallel import pooled_means_sharded
                pooled_all = pooled_means_sharded(self.llm.forward_segments, list(emb.split(lens)), self._event_rank,
                                                  self._event_world, self._event_group)
            else:
                pooled_all = self.llm.forward_segments(emb, lens)
        pooled_all = self._bf16_round(pooled_all)  # mean of bf16 is bf16
        qvec, pooled = pooled_all[K], pooled_all[:K]
        cos = self._bf16_round(ops.cosine(qvec.contiguous(), pooled.contiguous())).cpu()
        assert cos.shape[0] == K
        self.last_debug.update(cosine_raw=cos.clone())
        if self.cosine_override is not None:  # test hook: pin the branch below with forced similarities
            cos = torch.tensor(self.cosine_override, dtype=torch.float32)
        thr =float(torch.tensor(COSINE_THRESHOLD).to(self.dtype)) if self.dtype == torch.bfloat16 else COSINE_THRESHOLD
        minor_clusters = set(torch.where(cos < thr)[0].tolist())
        self.last_debug.update(assign=assign_h, cosine=cos, picked=sorted(picked_set))
        return [i for i in range(T) if assign_h[i] in minor_clusters and i not in picked_set]

    def compress_unimportant_events(self, mm_features: torch.Tensor, patch_num: int, minor_frame_indices: List[int]):
        """:434-447 (returns a copy, like the reference's .clone())"""
        total = mm_features.shape[0]
        if total % patch_num != 0:
            raise ValueError(f"total patches ({total}) not divisible by patches per frame ({patch_num})")
        out = mm_features.clone()
        if minor_frame_indices:
            ops.frame_mean_to_slot0(out, patch_num, torch.tensor(minor_frame_indices, dtype=torch.int32, device=out.device))
        return out

    def _get_compression_mask(self, pixel_values, batched_num_patches, grid_sizes, merge_sizes, modals,
                              threshold: float = 0.1, min_tokens: int = 1,
                              minor_frame_indices: Optional[List[int]] = None) -> torch.Tensor:
        """:383-432 -> bool [M] on the device"""
        minor = set(minor_frame_indices or [])
        masks = []
        row = 0
        gcount = 0
        for n, gs, ms, modal in zip(batched_num_patches.tolist(), grid_sizes.tolist(), merge_sizes.tolist(), modals):
            t, h, w = gs
            npatch = t * h * w
            if modal == "image" or (modal == "video" and t == 1):
                masks.append(torch.ones(n, dtype=torch.uint8, device=pixel_values.device))
            elif modal == "video":
                flags = torch.tensor([1 if gcount + f in minor else 0 for f in range(t)], dtype=torch.uint8,
                                     device=pixel_values.device)
                masks.append(ops.pixdiff_mask(pixel_values[row:row + npatch], t, (h // ms) * (w // ms), threshold,
                                              min_tokens, flags))
            else:
                masks.append(torch.ones(0, dtype=torch.uint8, device=pixel_values.device))
            row += npatch
            gcount += t
        return torch.cat(masks).bool()

    def _maybe_truncate_visual_tokens(self, mm_features, compression_mask, batched_num_patches, modals, input_ids,
                                      position_ids=None):
        """:349-381 -- packed rows only (position_ids given, i.e. training-style packing; inference passes None and gets
        its inputs back): when a sample's text was cut and fewer <image> placeholders survive than it has visual tokens,
        the surplus tokens (and their mask entries) are dropped. Samples are delimited by position_ids == 0."""
        ids = input_ids.reshape(-1).cpu()
        if position_ids is None or mm_features.shape[0] == int((ids == self.config.image_token_index).sum()):
            return mm_features, compression_mask
        pos = position_ids.reshape(-1).cpu()
        ends = [int(i) for i in torch.nonzero(pos == 0)[:, 0].tolist() if i > 0] + [int(ids.numel())]
        starts = [0] + ends[:-1]
        counts = [int((ids[a:b] == self.config.image_token_index).sum()) for a, b in zip(starts, ends)]
        keep = []
        for n_patches, modal in zip([int(x) for x in batched_num_patches.tolist()], modals):
            keep.append(torch.ones(0 if modal == "text" else n_patches, dtype=torch.bool))
        for n, m in zip(counts, keep):          # zip stops at the shorter list, exactly like the reference (:376-378)
            if m.numel() > 0:
                m[n:] = False
        keep = torch.cat(keep).to(mm_features.device)
        return mm_features[keep], compression_mask[keep.to(compression_mask.device)]

    def _compress_visual_tokens(self, compression_mask, input_ids, attention_mask):
        """:449-476 (inference subset). Returns (row index of every kept visual token, input_ids', mask')"""
        keep = compression_mask.cpu().numpy().astype(bool)
        ids = input_ids.cpu().numpy()
        sel = ids == self.config.image_token_index
        text_mask = ~sel
        text_mask[sel] = keep
        new_ids = torch.from_numpy(ids[text_mask])
        new_mask = attention_mask.cpu()[torch.from_numpy(text_mask)] if attention_mask is not None else None
        return torch.from_numpy(np.nonzero(keep)[0]), new_ids, new_mask

    # ------------------------------------------------------------------ multimodal assembly
    def prepare_inputs_labels_for_multimodal(self, input_ids=None, attention_mask=None, pixel_values=None,
                                             grid_sizes=None, merge_sizes=None, modals=None, total_image_num=0,
                                             if_visual=True, video_keys=None, position_ids=None):
        """:513-584 -> (inputs_embeds [1,S',H] on the device, attention_mask [1,S'])"""
        B, N = input_ids.shape
        assert B == 1, "Token compression is only supported for batch_size=1"
        ids = input_ids.reshape(-1).cpu()
        am = attention_mask.reshape(-1) if attention_mask is not None else None
        mm = None
        if if_visual:
            pixel_values = pixel_values.to(self.device)
            batched = grid_sizes.prod(dim=1).div(merge_sizes ** 2).long()
            with self._stage("encode"):
                mm = self.encode_images(pixel_values, grid_sizes, merge_sizes, video_keys=video_keys)
            text_rows = [m == "text" for m in modals]
            if any(text_rows):  # _get_valid_visual_tokens (:336-347)
                keep = torch.cat([torch.full((int(n),), not tr, dtype=torch.bool) for n, tr in zip(batched, text_rows)])
                mm = mm[keep.to(mm.device)].contiguous()
            assert mm.shape[0] % total_image_num == 0, f"{mm.shape[0]} % {total_image_num} != 0"
            frame_indices = self.select_events_based_on_summary(mm, total_image_num, self.all_timestamps)
            if self._is_helper():       # frame-sharded run: this rank's part (encode, gather, event passes) is done
                return None, None
            mm = self.compress_unimportant_events(mm, mm.shape[0] // total_image_num, frame_indices)
            mask = self._get_compression_mask(pixel_values, batched, grid_sizes, merge_sizes, modals,
                                              minor_frame_indices=frame_indices)
            mm, mask = self._maybe_truncate_visual_tokens(mm, mask, batched, modals, ids, position_ids)      # :555-562
            self.last_debug.update(minor_frames=frame_indices, compression_mask=mask)
            if self.use_token_compression:
                rows, ids, am = self._compress_visual_tokens(mask, ids, am)
            else:
                rows = torch.arange(mm.shape[0])
            idx = ids.clone().to(torch.int64)
            sel = idx == self.config.image_token_index
            assert int(sel.sum()) == rows.numel(), (int(sel.sum()), rows.numel())
            idx[sel] = -(rows.to(torch.int64) + 1)
        else:
            if self._is_helper():
                return None, None
            idx = ids.to(torch.int64)
        embeds = ops.gather_rows(self.llm.packed.embed, mm, idx.to(self.device))
        self.last_debug.update(input_ids=ids)
        return embeds.unsqueeze(0), (am.reshape(1, -1) if am is not None else None)

    # ------------------------------------------------------------------ text side
    def prepare_inputs(self, selection_module_output: str, original_text: Optional[str] = None):
        """:478-511 (inference branch) -> (tokenized new prompt, if_visual)"""
        if_visual, sel = parse_selection(selection_module_output)
        hist_qs = [self.hist_qs[i] for i in sel if i < len(self.hist_qs)]
        hist_as = [self.hist_as[i] for i in sel if i < len(self.hist_qs)]  # sic: bound by len(hist_qs) (:502)
        prompt = process_input_ids(original_text, if_visual, hist_qs, hist_as, self.current_question, self.tokenizer)
        return self.tokenizer(prompt, padding=False, return_tensors="pt"), if_visual

    def generate_language_module(self, input_ids=None, attention_mask=None, max_new_tokens: int = 50,
                                 do_sample: bool = False, allowed_ids: Optional[Sequence[int]] = None,
                                 eos_token_id=151645, **kw) -> torch.Tensor:
        """:877-908 -- greedy decode from token ids; returns prompt + new ids [1, S+n] (ids were passed, so HF
        returns the prompt too and the repetition penalty sees the prompt ids)"""
        ids = input_ids.reshape(-1).to(torch.int64)
        emb = self.llm.embed_tokens(ids)
        eos = [eos_token_id] if isinstance(eos_token_id, int) else list(eos_token_id)
        new = self.llm.generate(emb, max_new_tokens=max_new_tokens, eos_token_id=eos, do_sample=do_sample,
                                repetition_penalty=self.generation_config.get("repetition_penalty", 1.0),
                                allowed_ids=allowed_ids, prompt_ids=ids, prefix=self._prefix_slot("selection"))
        return torch.cat([ids, torch.tensor(new, dtype=torch.int64)]).unsqueeze(0)

    def qa_selection(self, current_question=None, hist_qs=None, hist_as=None, tokenizer=None, original_text=None,
                     input_ids=None, attention_mask=None, mode="FCC", select_gt=None, if_visual=None, **kwargs):
        """:809-865"""
        self.tokenizer, self.hist_qs, self.hist_as = tokenizer, hist_qs, hist_as
        self.current_question = current_question
        self.all_timestamps = torch.tensor(kwargs.pop("all_timestamps", None))
        new_ids, new_mask, out_str, vis = input_ids, attention_mask, "", True
        if mode == "FCC":
            if len(hist_qs) > 0:
                if self._shard is None:
                    out_str = select_qas(current_question, hist_qs, hist_as, self, tokenizer)
                else:        # frame-sharded run: one retrieval decode on the llm_rank, the string goes to every rank
                    out_str = self._bcast(None if self._is_helper() else
                                          select_qas(current_question, hist_qs, hist_as, self, tokenizer))
        elif mode == "AC":
            pass
        elif mode == "NC":
            if len(hist_qs) > 0:
                out_str = "[yes]"
        elif mode == "gt":
            assert select_gt is not None, "in gt mode, you should provide selection gt"
            if len(hist_qs) > 0:
                out_str = "[" + ",".join(["yes" if if_visual else "no"] + [str(n) for n in select_gt]) + "]"
        else:
            raise ValueError(f"unknown mode {mode}")
        if out_str:
            new_inputs, vis = self.prepare_inputs(out_str, original_text=original_text)
            new_ids, new_mask = new_inputs["input_ids"], new_inputs["attention_mask"]
        return {"new_input_ids": new_ids, "new_attention_mask": new_mask, "selection_module_output": out_str,
                "input_ids": input_ids, "attention_mask": attention_mask, "if_visual": vis, **kwargs}

    @torch.no_grad()
    def generate(self, pixel_values=None, grid_sizes=None, merge_sizes=None, modals=None, new_input_ids=None,
                 new_attention_mask=None, selection_module_output="", if_visual=True, video_keys=None, **kwargs):
        """:753-807 -> (new token ids [1, n], selection_module_output)"""
        for k in ("input_ids", "past_key_values", "attention_mask", "position_ids", "tokenizer", "hist_qs", "hist_as",
                  "current_question", "original_text"):
            kwargs.pop(k, None)
        total_image_num = kwargs.pop("total_image_num", None)
        if "inputs_embeds" in kwargs:
            raise NotImplementedError("`inputs_embeds` is not supported")
        helper = self._is_helper()
        if pixel_values is not None:
            embeds, _ = self.prepare_inputs_labels_for_multimodal(
                input_ids=new_input_ids, attention_mask=new_attention_mask, pixel_values=pixel_values,
                grid_sizes=grid_sizes, merge_sizes=merge_sizes, modals=modals, total_image_num=total_image_num,
                if_visual=if_visual, video_keys=video_keys)
            embeds = None if helper else embeds[0]
        elif not helper:
            embeds = self.llm.embed_tokens(new_input_ids.reshape(-1))
        if helper:     # frame-sharded run: the llm_rank prefills and decodes; its new token ids arrive by broadcast
            got = self._bcast(None)
            return torch.tensor(got, dtype=torch.int64).reshape(1, len(got)), selection_module_output
        g = dict(self.generation_config)
        g.update({k: v for k, v in kwargs.items() if k in ("do_sample", "temperature", "top_k", "top_p",
                                                            "repetition_penalty", "eos_token_id", "generator",
                                                            "sampler", "seed")})
        eos = g.get("eos_token_id", [])
        eos = [eos] if isinstance(eos, int) else list(eos)
        new = self.llm.generate(embeds, max_new_tokens=int(kwargs.get("max_new_tokens", 1024)), eos_token_id=eos,
                                stage_times=self.stage_times,
                                do_sample=bool(g.get("do_sample", False)), temperature=float(g.get("temperature", 1.0)),
                                top_k=int(g.get("top_k", 0) or 0), top_p=float(g.get("top_p", 1.0)),
                                repetition_penalty=float(g.get("repetition_penalty", 1.0)), generator=g.get("generator"),
                                sampler=g.get("sampler", "device"), seed=g.get("seed"),
                                prefix=self._prefix_slot("answer"))
        if self._shard is not None:
            self._bcast(list(new))
        return torch.tensor(new, dtype=torch.int64).unsqueeze(0), selection_module_output
