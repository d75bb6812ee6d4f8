"""Checkpoint / config loading (cogstream_amd/checkpoint.py) without real weights: the reference repository ships
model.safetensors.index.json but not the shards (git-LFS), so shards are SYNTHESISED from the index's own 779 names
and shard assignment at scaled-down dimensions; names, byte total and configs are pinned by
tests/golden/checkpoint_index.json (the reference's metadata files, data)."""
import json
import os

import pytest
import torch

G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def meta():
    return json.load(open(os.path.join(G, "checkpoint_index.json")))


def _write_dir(tmp, meta, files=("config.json", "generation_config.json", "preprocessor_config.json", "processor_config.json")):
    for n in files:
        with open(os.path.join(tmp, n), "w") as f:
            json.dump(meta[n], f)


def test_configs_and_tensor_inventory_match_the_reference_checkpoint(meta, tmp_path):
    from cogstream_amd import checkpoint as ck
    from cogstream_amd.chat import DEFAULT_GENERATION
    from cogstream_amd.weights import LlmConfig, VisionConfig
    _write_dir(str(tmp_path), meta)
    c = ck.load_configs(str(tmp_path))
    assert c["vision"] == VisionConfig() and c["llm"] == LlmConfig()          # the defaults ARE the shipped dimensions
    assert c["generation"] == {k: v for k, v in DEFAULT_GENERATION.items()} | {"bos_token_id": 151643}
    assert c["use_token_compression"] is True and c["torch_dtype"] == "bfloat16"
    assert (c["processor"]["max_tokens"], c["processor"]["min_tokens"], c["processor"]["video_merge_size"]) == (16384, 16, 2)
    want = ck.expected_tensors(c["vision"], c["llm"])
    assert len(want) == 779 and set(want) == set(meta["weight_map"])
    nbytes = sum(2 * int(torch.Size(s).numel()) for s in want.values())
    assert nbytes == meta["total_size"] == 16089489888                        # model.safetensors.index.json metadata


def _small_cfgs():
    from cogstream_amd.weights import LlmConfig, VisionConfig
    # the reference's layer counts (27 / 28: the same 779 names), small widths
    return (VisionConfig(hidden_size=64, intermediate_size=72, num_hidden_layers=27, num_attention_heads=2),
            LlmConfig(hidden_size=128, intermediate_size=192, num_hidden_layers=28, num_attention_heads=2,
                      num_key_value_heads=1, vocab_size=320, image_token_index=300, eos_token_id=299))


def _synth(tmp, meta, drop=None, extra=None):
    """shards with the reference's names and shard assignment, random values at the small dimensions"""
    from safetensors.torch import save_file
    from cogstream_amd import checkpoint as ck
    vc, lc = _small_cfgs()
    want = ck.expected_tensors(vc, lc)
    g = torch.Generator().manual_seed(5)
    full = {n: (torch.randn(*s, generator=g) * 0.05).bfloat16() for n, s in want.items()}
    wm = dict(meta["weight_map"])
    if drop:
        wm.pop(drop)
    if extra:
        wm[extra] = "model-00001-of-00004.safetensors"
        full[extra] = torch.zeros(3).bfloat16()
    by_file = {}
    for n, fn in wm.items():
        by_file.setdefault(fn, {})[n] = full[n]
    for fn, ts in by_file.items():
        save_file(ts, os.path.join(tmp, fn))
    json.dump({"metadata": {"total_size": 0}, "weight_map": wm}, open(os.path.join(tmp, ck.INDEX), "w"))
    return full, vc, lc


def test_streamed_load_consumes_every_tensor_once_and_packs_identically(meta, tmp_path):
    from cogstream_amd import checkpoint as ck
    from cogstream_amd.weights import PackedLlm, PackedProjector, PackedVit
    tmp = str(tmp_path)
    full, vc, lc = _synth(tmp, meta)
    reader = ck.Checkpoint(tmp, device="cpu")
    assert len(reader.names()) == 779 and len(set(reader.weight_map.values())) == 4
    vit_v, proj_v, llm_v = ck.state_views(reader)
    a = (PackedVit(vit_v, vc, torch.bfloat16, "cpu"), PackedProjector(proj_v, torch.bfloat16, "cpu"),
         PackedLlm(llm_v, lc, torch.bfloat16, "cpu"))
    reader.check_consumed()                                        # 779 reads, each exactly once
    assert sum(reader.reads.values()) == 779 and set(reader.reads.values()) == {1}
    sub = lambda pre, top=(): {(k if k in top else k[len(pre):]): v for k, v in full.items()
                               if (k in top) or (k.startswith(pre) and not (pre == ck.LLM_PREFIX and (k.startswith(ck.VIT_PREFIX) or k.startswith(ck.PROJ_PREFIX))))}
    b = (PackedVit(sub(ck.VIT_PREFIX), vc, torch.bfloat16, "cpu"), PackedProjector(sub(ck.PROJ_PREFIX), torch.bfloat16, "cpu"),
         PackedLlm(sub(ck.LLM_PREFIX, ("lm_head.weight",)), lc, torch.bfloat16, "cpu"))
    for x, y in ((a[0], b[0]), (a[2], b[2])):
        assert len(x.keep) == len(y.keep) and all(torch.equal(p, q) for p, q in zip(x.keep, y.keep))
    assert all(torch.equal(getattr(a[1], n), getattr(b[1], n)) for n in ("w1", "b1", "w2", "b2"))
    assert len(list(llm_v)) == 28 * 12 + 3 and len(vit_v) == 27 * 16 + 4 and len(proj_v) == 4


def test_missing_shard_missing_tensor_and_leftover_tensor_fail_loudly(meta, tmp_path):
    from cogstream_amd import checkpoint as ck
    from cogstream_amd.weights import PackedVit
    d1, d2, d3 = (str(tmp_path / n) for n in ("a", "b", "c"))
    for d in (d1, d2, d3):
        os.makedirs(d)
    json.dump({"metadata": {}, "weight_map": meta["weight_map"]}, open(os.path.join(d1, ck.INDEX), "w"))
    with pytest.raises(FileNotFoundError, match="shard files"):            # the reference repo's own state: index, no shards
        ck.Checkpoint(d1)
    _, vc, lc = _synth(d2, meta, drop="model.vision_encoder.encoder.layers.3.mlp.fc1.bias")
    r = ck.Checkpoint(d2)
    with pytest.raises(KeyError):
        PackedVit(ck.state_views(r)[0], vc, torch.bfloat16, "cpu")
    _synth(d3, meta, extra="model.layers.0.self_attn.extra.weight")
    r = ck.Checkpoint(d3)
    for n in list(meta["weight_map"]):
        r.tensor(n)
    with pytest.raises(RuntimeError, match="not consumed"):
        r.check_consumed()


def test_save_checkpoint_round_trip_and_adapter_dir(tmp_path):
    from safetensors.torch import save_file
    from cogstream_amd import checkpoint as ck
    from cogstream_amd.weights import LlmConfig, VisionConfig, random_llm_state, random_lora_state, random_proj_state, random_vit_state
    vc = VisionConfig(hidden_size=64, intermediate_size=72, num_hidden_layers=2, num_attention_heads=2)
    lc = LlmConfig(hidden_size=128, intermediate_size=192, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=1,
                   vocab_size=320, image_token_index=300, eos_token_id=299)
    vs, ps, ls = random_vit_state(vc), random_proj_state(64, 128), random_llm_state(lc)
    d = str(tmp_path / "m")
    ck.save_checkpoint(d, vs, ps, ls, vc, lc, generation={"do_sample": False, "eos_token_id": [299]}, n_shards=3)
    c = ck.load_configs(d)
    assert c["vision"] == vc and c["llm"] == lc and c["generation"] == {"do_sample": False, "eos_token_id": [299]}
    r = ck.Checkpoint(d)
    assert set(r.names()) == set(ck.expected_tensors(vc, lc))
    v, p, l = ck.state_views(r)
    assert torch.equal(v["post_layernorm.weight"], vs["post_layernorm.weight"].bfloat16())
    assert torch.equal(l["lm_head.weight"], ls["lm_head.weight"].bfloat16()) and "lm_head.weight" in l and "nope" not in l
    ad = str(tmp_path / "adapter")
    os.makedirs(ad)
    lora = random_lora_state(lc, r=4)
    save_file({k: v.contiguous() for k, v in lora.items()}, os.path.join(ad, "adapter_model.safetensors"))
    json.dump({"r": 4, "lora_alpha": 8}, open(os.path.join(ad, "adapter_config.json"), "w"))
    st, alpha = ck.load_adapter_state(ad)
    assert alpha == 8.0 and set(st) == set(lora)


def test_tied_embeddings_load_and_wrong_shapes_fail_by_name(tmp_path):
    """tie_word_embeddings: lm_head.weight is an alias of model.embed_tokens.weight -- PackedLlm reads both names, which
    is one genuine read plus one aliased read, not 'read more than once'. A shard whose tensor shapes do not match
    config.json fails in check_shapes with the tensor's name, before any packing."""
    from safetensors.torch import save_file
    from cogstream_amd import checkpoint as ck
    from cogstream_amd.weights import LlmConfig, PackedLlm, VisionConfig, random_llm_state, random_proj_state, random_vit_state
    vc = VisionConfig(hidden_size=64, intermediate_size=72, num_hidden_layers=1, num_attention_heads=2)
    lc = LlmConfig(hidden_size=128, intermediate_size=192, num_hidden_layers=1, num_attention_heads=2, num_key_value_heads=1,
                   vocab_size=320, image_token_index=300, eos_token_id=299)
    ls = random_llm_state(lc)
    d = str(tmp_path / "tied")
    ck.save_checkpoint(d, random_vit_state(vc), random_proj_state(64, 128), ls, vc, lc, n_shards=1)
    # make it a tied checkpoint: drop lm_head.weight from the shard and the index, flag it in config.json
    fn = os.path.join(d, "model-00001-of-00001.safetensors")
    from safetensors import safe_open
    with safe_open(fn, framework="pt") as f:
        ts = {k: f.get_tensor(k) for k in f.keys() if k != "lm_head.weight"}
    save_file(ts, fn)
    idx = json.load(open(os.path.join(d, ck.INDEX)))
    del idx["weight_map"]["lm_head.weight"]
    json.dump(idx, open(os.path.join(d, ck.INDEX), "w"))
    cfg = json.load(open(os.path.join(d, "config.json")))
    cfg["tie_word_embeddings"] = True
    json.dump(cfg, open(os.path.join(d, "config.json"), "w"))
    c = ck.load_configs(d)
    assert c["tie_word_embeddings"] is True
    r = ck.Checkpoint(d)
    want = ck.expected_tensors(c["vision"], c["llm"], True)
    assert "lm_head.weight" not in want and set(want) == set(r.names())
    r.check_shapes(want)
    _, _, llm_v = ck.state_views(r, True)
    packed = PackedLlm(llm_v, lc, torch.bfloat16, "cpu")
    assert torch.equal(packed.lm_head, packed.embed)                 # the same values, read through the alias
    assert r.reads["model.embed_tokens.weight"] == 1 and r.alias_reads["model.embed_tokens.weight"] == 1
    for n in r.names():                                               # the other two sub-trees, then nothing may be left
        if n.startswith(ck.VIT_PREFIX) or n.startswith(ck.PROJ_PREFIX):
            r.tensor(n)
    r.check_consumed()
    # wrong-size shard: config.json says hidden 128, the tensors say 64
    lc_bad = LlmConfig(hidden_size=64, intermediate_size=192, num_hidden_layers=1, num_attention_heads=2, num_key_value_heads=1,
                       vocab_size=320, image_token_index=300, eos_token_id=299)
    d2 = str(tmp_path / "bad")
    ck.save_checkpoint(d2, random_vit_state(vc), random_proj_state(64, 64), random_llm_state(lc_bad), vc, lc, n_shards=1)
    r2 = ck.Checkpoint(d2)
    with pytest.raises(RuntimeError, match="embed_tokens|readout|layers"):
        r2.check_shapes(ck.expected_tensors(vc, lc))


@pytest.mark.skipif(not os.path.isdir("/root/reference/model"), reason="build container only")
def test_processor_from_pretrained_on_the_reference_directory():
    """AutoProcessor.from_pretrained equivalent on the reference's own model directory (tokenizer + processor configs
    are present there; weights are not) -- and the model loader names the missing shards instead of crashing"""
    import numpy as np
    from cogstream_amd import checkpoint as ck
    from cogstream_amd.processing import CogStreamProcessor, synthetic_clip
    proc = CogStreamProcessor.from_pretrained("/root/reference/model")
    assert (proc.max_tokens, proc.min_tokens, proc.video_merge_size) == (16384, 16, 2)
    frames, ts = synthetic_clip(8, 224, 224)
    out = proc(conversation=[{"role": "user", "content": [{"type": "video", "video": frames, "timestamps": ts},
                                                          {"type": "text", "text": "What is happening in the video?"}]}],
               add_system_prompt=True, add_generation_prompt=True)
    assert out["input_ids"].shape == (1, 621) and int((out["input_ids"] == 151665).sum()) == 512
    with pytest.raises(FileNotFoundError, match="git-LFS"):
        ck.Checkpoint("/root/reference/model")
