"""Checkpoint / config loading: the directory the reference is instantiated from.

    AutoModelForCausalLM.from_pretrained(model_path, trust_remote_code=True, torch_dtype=torch.bfloat16,
                                         attn_implementation="flash_attention_2")      evaluate/answer_generate.py:173-178
    AutoProcessor.from_pretrained(model_path, trust_remote_code=True)                  :179
    PeftModel.from_pretrained(model, path, adapter_name=...), model.load_adapter(...)  :181-182

reads, from that directory: config.json (dimensions, image_token_index, use_token_compression; model/config.json),
generation_config.json (the sampling defaults generate() runs with), preprocessor_config.json / processor_config.json
(token budget, merge size), model.safetensors.index.json + its shards (779 tensors, module tree of
model/cogreasoner_chat.py:214-248,591-598), and the tokenizer files (through transformers' Qwen2 tokenizer, host
side, as the reference does). Tensors are streamed shard -> HBM one at a time with safetensors.safe_open(device=...)
and packed into the kernels' layouts as they arrive (cogstream_amd.weights.Packed*); nothing is staged as a whole
state dict on the host. Every tensor of the index must be consumed exactly once and none may be missing."""
from __future__ import annotations

import json
import os
from collections import Counter
from collections.abc import Mapping
from typing import Dict, Iterable, List, Optional, Tuple

import torch

from .weights import LlmConfig, VisionConfig

INDEX = "model.safetensors.index.json"
SINGLE = "model.safetensors"
VIT_PREFIX = "model.vision_encoder."
PROJ_PREFIX = "model.mm_projector."
LLM_PREFIX = "model."


def _read_json(path: str, name: str, required: bool = True) -> dict:
    p = os.path.join(path, name)
    if not os.path.exists(p):
        if required:
            raise FileNotFoundError(f"{p}: not a CogReasoner checkpoint directory (no {name})")
        return {}
    with open(p, encoding="utf-8") as f:
        return json.load(f)


def load_configs(path: str) -> Dict[str, object]:
    """-> dict(vision=VisionConfig, llm=LlmConfig, generation=dict, use_token_compression=bool, processor=dict,
    torch_dtype=str, tie_word_embeddings=bool) from the checkpoint directory's JSON files"""
    c = _read_json(path, "config.json")
    if c.get("mm_projector_type", "mlp2x_gelu") != "mlp2x_gelu":
        raise ValueError(f"mm_projector_type {c['mm_projector_type']!r}: only the reference's mlp2x_gelu projector "
                         "(model/cogreasoner_chat.py:179-211) is implemented")
    if c.get("rope_scaling") or c.get("use_sliding_window"):
        raise ValueError("rope_scaling / sliding-window attention are not used by the reference checkpoint and not implemented")
    v = c.get("vision_encoder_config") or {}
    vision = VisionConfig(hidden_size=v.get("hidden_size", 1152), intermediate_size=v.get("intermediate_size", 4304),
                          num_hidden_layers=v.get("num_hidden_layers", 27), num_attention_heads=v.get("num_attention_heads", 16),
                          num_channels=v.get("num_channels", 3), patch_size=v.get("patch_size", 14),
                          layer_norm_eps=v.get("layer_norm_eps", 1e-6))
    eos = c.get("eos_token_id", 151645)
    llm = LlmConfig(hidden_size=c["hidden_size"], intermediate_size=c["intermediate_size"],
                    num_hidden_layers=c["num_hidden_layers"], num_attention_heads=c["num_attention_heads"],
                    num_key_value_heads=c.get("num_key_value_heads", c["num_attention_heads"]), vocab_size=c["vocab_size"],
                    rms_norm_eps=c.get("rms_norm_eps", 1e-6), rope_theta=c.get("rope_theta", 1e6),
                    image_token_index=c.get("image_token_index", 151665),
                    eos_token_id=eos[0] if isinstance(eos, (list, tuple)) else eos)
    gen = _read_json(path, "generation_config.json", required=False)
    gen = {k: v for k, v in gen.items() if k in ("do_sample", "temperature", "top_k", "top_p", "repetition_penalty",
                                                 "eos_token_id", "pad_token_id", "bos_token_id")}
    pre = _read_json(path, "preprocessor_config.json", required=False)
    prc = _read_json(path, "processor_config.json", required=False)
    processor = {"max_tokens": pre.get("max_tokens", 16384), "min_tokens": pre.get("min_tokens", 16),
                 "patch_size": pre.get("patch_size", 14), "image_mean": pre.get("image_mean", [0.5, 0.5, 0.5]),
                 "image_std": pre.get("image_std", [0.5, 0.5, 0.5]), "rescale_factor": pre.get("rescale_factor", 1 / 255),
                 "resample": pre.get("resample", 3), "video_merge_size": prc.get("video_merge_size", 2),
                 "image_merge_size": prc.get("image_merge_size", 1), "fps": prc.get("fps", 1),
                 "max_frames": prc.get("max_frames", 128)}
    return {"vision": vision, "llm": llm, "generation": gen, "use_token_compression": bool(c.get("use_token_compression", True)),
            "processor": processor, "torch_dtype": c.get("torch_dtype", "bfloat16"),
            "tie_word_embeddings": bool(c.get("tie_word_embeddings", False))}


def expected_tensors(vision: VisionConfig, llm: LlmConfig, tie_word_embeddings: bool = False) -> Dict[str, Tuple[int, ...]]:
    """name -> shape of every tensor the module tree holds (model/cogreasoner_chat.py:214-248,591-598; the names of
    model/model.safetensors.index.json). At the shipped dimensions: 779 tensors, 16 089 489 888 bytes in bf16."""
    H, I = vision.hidden_size, vision.intermediate_size
    out: Dict[str, Tuple[int, ...]] = {
        VIT_PREFIX + "embeddings.patch_embedding.weight": (H, vision.num_channels, vision.patch_size, vision.patch_size),
        VIT_PREFIX + "embeddings.patch_embedding.bias": (H,),
        VIT_PREFIX + "post_layernorm.weight": (H,), VIT_PREFIX + "post_layernorm.bias": (H,),
    }
    for i in range(vision.num_hidden_layers):
        p = f"{VIT_PREFIX}encoder.layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            out[p + f"self_attn.{n}.weight"], out[p + f"self_attn.{n}.bias"] = (H, H), (H,)
        for n in ("layer_norm1", "layer_norm2"):
            out[p + n + ".weight"], out[p + n + ".bias"] = (H,), (H,)
        out[p + "mlp.fc1.weight"], out[p + "mlp.fc1.bias"] = (I, H), (I,)
        out[p + "mlp.fc2.weight"], out[p + "mlp.fc2.bias"] = (H, I), (H,)
    D = llm.hidden_size
    out[PROJ_PREFIX + "readout.0.weight"], out[PROJ_PREFIX + "readout.0.bias"] = (D, H), (D,)
    out[PROJ_PREFIX + "readout.2.weight"], out[PROJ_PREFIX + "readout.2.bias"] = (D, D), (D,)
    kvd, qd, F = llm.num_key_value_heads * llm.head_dim, llm.num_attention_heads * llm.head_dim, llm.intermediate_size
    out[LLM_PREFIX + "embed_tokens.weight"] = (llm.vocab_size, D)
    out[LLM_PREFIX + "norm.weight"] = (D,)
    if not tie_word_embeddings:
        out["lm_head.weight"] = (llm.vocab_size, D)
    for i in range(llm.num_hidden_layers):
        p = f"{LLM_PREFIX}layers.{i}."
        out[p + "input_layernorm.weight"] = out[p + "post_attention_layernorm.weight"] = (D,)
        out[p + "self_attn.q_proj.weight"], out[p + "self_attn.q_proj.bias"] = (qd, D), (qd,)
        out[p + "self_attn.k_proj.weight"], out[p + "self_attn.k_proj.bias"] = (kvd, D), (kvd,)
        out[p + "self_attn.v_proj.weight"], out[p + "self_attn.v_proj.bias"] = (kvd, D), (kvd,)
        out[p + "self_attn.o_proj.weight"] = (D, qd)
        out[p + "mlp.gate_proj.weight"] = out[p + "mlp.up_proj.weight"] = (F, D)
        out[p + "mlp.down_proj.weight"] = (D, F)
    return out


class Checkpoint:
    """lazy reader over model.safetensors.index.json + shards (or one model.safetensors); counts every read"""

    def __init__(self, path: str, device="cpu"):
        from safetensors import safe_open
        self.path, self.device = path, str(torch.device(device))
        self._open = safe_open
        self._files: Dict[str, object] = {}
        idx = os.path.join(path, INDEX)
        if os.path.exists(idx):
            with open(idx, encoding="utf-8") as f:
                d = json.load(f)
            self.weight_map: Dict[str, str] = dict(d["weight_map"])
            self.total_size = d.get("metadata", {}).get("total_size")
        elif os.path.exists(os.path.join(path, SINGLE)):
            with safe_open(os.path.join(path, SINGLE), framework="pt", device="cpu") as f:
                self.weight_map = {k: SINGLE for k in f.keys()}
            self.total_size = None
        else:
            raise FileNotFoundError(f"{path}: neither {INDEX} nor {SINGLE}")
        missing = sorted({fn for fn in self.weight_map.values() if not os.path.exists(os.path.join(path, fn))})
        if missing:
            raise FileNotFoundError(f"{path}: shard files named by the index are missing: {missing} "
                                    "(the reference repository ships the index only; the weights are git-LFS objects)")
        self.reads: Counter = Counter()
        self.alias_reads: Counter = Counter()    # reads through an alias (tied lm_head -> embed_tokens): not "twice"

    def names(self) -> List[str]:
        return list(self.weight_map)

    def _file(self, name: str):
        fn = self.weight_map[name]          # KeyError = the checkpoint lacks a tensor the module tree needs
        f = self._files.get(fn)
        if f is None:
            f = self._files[fn] = self._open(os.path.join(self.path, fn), framework="pt", device=self.device)
        return f

    def tensor(self, name: str, aliased: bool = False) -> torch.Tensor:
        f = self._file(name)
        (self.alias_reads if aliased else self.reads)[name] += 1
        return f.get_tensor(name)

    def shape(self, name: str) -> Tuple[int, ...]:
        """shape from the shard's header (no tensor data is read)"""
        return tuple(self._file(name).get_slice(name).get_shape())

    def check_shapes(self, expected: Dict[str, Tuple[int, ...]]) -> None:
        """every tensor of the module tree has the shape config.json implies -- a shard of another model size fails
        here, by name, instead of somewhere inside the packers or the kernels"""
        bad = [(n, self.shape(n), tuple(s)) for n, s in expected.items() if n in self.weight_map and self.shape(n) != tuple(s)]
        if bad:
            raise RuntimeError(f"{self.path}: {len(bad)} tensors do not have the shape config.json implies, e.g. "
                               + "; ".join(f"{n}: {got} != {want}" for n, got, want in bad[:4]))

    def close(self):
        self._files.clear()

    def check_consumed(self, extra_ok: Iterable[str] = ()) -> None:
        """every tensor read exactly once, nothing left over (rotary inv_freq buffers of older exports are ignored)"""
        ok = set(extra_ok)
        left = [n for n in self.weight_map if self.reads[n] == 0 and n not in ok and "rotary_emb.inv_freq" not in n]
        twice = [n for n, c in self.reads.items() if c > 1]
        if left or twice:
            raise RuntimeError(f"checkpoint tensors not consumed: {left[:8]}{'...' if len(left) > 8 else ''}; "
                               f"read more than once: {twice[:8]}")


class StateView(Mapping):
    """the slice of a Checkpoint one sub-module sees, under the names cogstream_amd.weights.Packed* ask for"""

    def __init__(self, ckpt: Checkpoint, prefix: str, top_level: Tuple[str, ...] = (), alias: Optional[Dict[str, str]] = None,
                 exclude: Tuple[str, ...] = ()):
        self.ckpt, self.prefix, self.top_level, self.alias = ckpt, prefix, set(top_level), dict(alias or {})
        self.exclude = tuple(exclude)     # sub-trees under `prefix` that belong to another view

    def _full(self, k: str) -> str:
        if k in self.alias:
            return self.alias[k]
        return k if k in self.top_level else self.prefix + k

    def __getitem__(self, k: str) -> torch.Tensor:
        return self.ckpt.tensor(self._full(k), aliased=k in self.alias)

    def __contains__(self, k) -> bool:
        return self._full(k) in self.ckpt.weight_map

    def __iter__(self):
        for n in self.ckpt.weight_map:
            if n in self.top_level:
                yield n
            elif n.startswith(self.prefix) and not n.startswith(self.exclude or ("\0",)):
                yield n[len(self.prefix):]

    def __len__(self) -> int:
        return sum(1 for _ in self)


def state_views(ckpt: Checkpoint, tie_word_embeddings: bool = False):
    """-> (vit, projector, llm) views keyed like weights.random_*_state"""
    alias = {"lm_head.weight": LLM_PREFIX + "embed_tokens.weight"} if tie_word_embeddings and "lm_head.weight" not in ckpt.weight_map else None
    return (StateView(ckpt, VIT_PREFIX), StateView(ckpt, PROJ_PREFIX),
            StateView(ckpt, LLM_PREFIX, top_level=("lm_head.weight",), alias=alias, exclude=(VIT_PREFIX, PROJ_PREFIX)))


def load_tokenizer(path: str):
    """the checkpoint's Qwen2 BPE tokenizer (vocab.json / merges.txt / added_tokens.json), through transformers --
    host-side third-party code, exactly what processor.tokenizer is in the reference (answer_generate.py:180)"""
    try:
        from transformers import Qwen2TokenizerFast as Tok
    except ImportError:                       # pragma: no cover
        from transformers import Qwen2Tokenizer as Tok
    return Tok.from_pretrained(path)


def load_adapter_state(path: str, device="cpu"):
    """peft adapter directory (adapter_config.json + adapter_model.safetensors | .bin) -> (state dict, lora_alpha)"""
    cfg = _read_json(path, "adapter_config.json", required=False)
    alpha = float(cfg.get("lora_alpha", 16.0))
    st = os.path.join(path, "adapter_model.safetensors")
    if os.path.exists(st):
        from safetensors.torch import load_file
        return load_file(st, device=str(torch.device(device))), alpha
    b = os.path.join(path, "adapter_model.bin")
    if os.path.exists(b):
        return torch.load(b, map_location=device, weights_only=True), alpha
    raise FileNotFoundError(f"{path}: no adapter_model.safetensors / adapter_model.bin")


def save_checkpoint(path: str, vit_state, proj_state, llm_state, vision: VisionConfig, llm: LlmConfig,
                    generation: Optional[dict] = None, n_shards: int = 2, dtype=torch.bfloat16,
                    use_token_compression: bool = True) -> None:
    """write a checkpoint directory in the reference's layout (config.json, generation_config.json,
    preprocessor/processor configs, index + shards). Used by the tests to synthesise checkpoints (no real weights are
    reachable: README.md:56-58) and handy for exporting random-init models at other sizes."""
    from safetensors.torch import save_file
    os.makedirs(path, exist_ok=True)
    full: Dict[str, torch.Tensor] = {}
    for k, v in vit_state.items():
        full[VIT_PREFIX + k] = v
    for k, v in proj_state.items():
        full[PROJ_PREFIX + k] = v
    for k, v in llm_state.items():
        full[k if k == "lm_head.weight" else LLM_PREFIX + k] = v
    names = sorted(full)
    per = -(-len(names) // n_shards)
    weight_map, total = {}, 0
    for s in range(n_shards):
        part = names[s * per:(s + 1) * per]
        if not part:
            continue
        fn = f"model-{s + 1:05d}-of-{n_shards:05d}.safetensors"
        tensors = {n: full[n].detach().to("cpu", dtype).contiguous() for n in part}
        save_file(tensors, os.path.join(path, fn))
        for n, t in tensors.items():
            weight_map[n] = fn
            total += t.numel() * t.element_size()
    with open(os.path.join(path, INDEX), "w") as f:
        json.dump({"metadata": {"total_size": total}, "weight_map": weight_map}, f)
    cfg = {"architectures": ["Videollama3Qwen2ForCausalLM"], "model_type": "videollama3_qwen2",
           # the reference's trust_remote_code entry points (model/config.json:6-9); cogstream_amd.auto resolves them
           "auto_map": {"AutoConfig": "configuration_videollama3.Videollama3Qwen2Config",
                        "AutoModelForCausalLM": "cogreasoner_chat.Videollama3Qwen2ForCausalLM"},
           "hidden_size": llm.hidden_size, "intermediate_size": llm.intermediate_size,
           "num_hidden_layers": llm.num_hidden_layers, "num_attention_heads": llm.num_attention_heads,
           "num_key_value_heads": llm.num_key_value_heads, "vocab_size": llm.vocab_size, "rms_norm_eps": llm.rms_norm_eps,
           "rope_theta": llm.rope_theta, "image_token_index": llm.image_token_index, "eos_token_id": llm.eos_token_id,
           "mm_projector_type": "mlp2x_gelu", "tie_word_embeddings": False, "use_token_compression": use_token_compression,
           "torch_dtype": str(dtype).replace("torch.", ""),
           "vision_encoder_config": {"hidden_size": vision.hidden_size, "intermediate_size": vision.intermediate_size,
                                     "num_hidden_layers": vision.num_hidden_layers,
                                     "num_attention_heads": vision.num_attention_heads, "patch_size": vision.patch_size,
                                     "model_type": "videollama3_vision_encoder"}}
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(cfg, f, indent=1)
    if generation is not None:
        with open(os.path.join(path, "generation_config.json"), "w") as f:
            json.dump(generation, f, indent=1)
    with open(os.path.join(path, "preprocessor_config.json"), "w") as f:     # model/preprocessor_config.json:2-5,10-26
        json.dump({"auto_map": {"AutoImageProcessor": "image_processing_videollama3.Videollama3ImageProcessor",
                                "AutoProcessor": "processing_cogreasoner.Videollama3Qwen2Processor"},
                   "image_mean": [0.5, 0.5, 0.5], "image_std": [0.5, 0.5, 0.5], "max_tokens": 16384, "min_tokens": 16,
                   "patch_size": 14, "resample": 3, "rescale_factor": 1 / 255}, f, indent=1)


def save_byte_tokenizer(path: str) -> None:
    """tokenizer files for SYNTHETIC checkpoints (tests, rehearsals): a genuine Qwen2TokenizerFast -- the class
    load_tokenizer() returns for the real checkpoint -- over a byte-level vocabulary without merges (ids 0..255 = UTF-8
    bytes) plus the chat specials at 256..259 (<|im_start|>, <|im_end|>, <image>, <|endoftext|>). The real vocab.json /
    merges.txt are reference data that does not ship with this repo."""
    from transformers import Qwen2TokenizerFast
    # GPT-2's printable stand-ins for the 256 byte values (what a byte-level BPE vocabulary is keyed by)
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(0xA1, 0xAC + 1)) + list(range(0xAE, 0xFF + 1))
    cs = bs[:]
    for b in range(256):
        if b not in bs:          # the 68 unprintable bytes map to code points 256, 257, ...
            cs.append(256 + len(bs) - 188)
            bs.append(b)
    vocab = {chr(c): b for b, c in zip(bs, cs)}
    specials = ["<|im_start|>", "<|im_end|>", "<image>", "<|endoftext|>"]
    for i, t in enumerate(specials):
        vocab[t] = 256 + i
    tok = Qwen2TokenizerFast(vocab=vocab, merges=[], unk_token=None, eos_token="<|im_end|>", pad_token="<|endoftext|>",
                             additional_special_tokens=specials)
    os.makedirs(path, exist_ok=True)
    tok.save_pretrained(path)
