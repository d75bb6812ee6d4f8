"""CPU-side checks: the C-ABI library loads and exports every symbol include/cogs.h declares (no compute
calls without a GPU), the product path refuses to run without a GPU, and the host preprocessing contract."""
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from cogstream_amd import _lib as L
    hdr = open(os.path.join(ROOT, "include", "cogs.h")).read()
    declared = set(re.findall(r"\b(cogs_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"cogs_status"}
    assert len(declared) >= 30
    for name in sorted(declared):
        assert hasattr(L.lib, name), f"{name} declared in include/cogs.h but not exported"
        assert name in L.SIGNATURES, f"{name} has no ctypes signature"
    assert L.lib.cogs_status_string(-4).decode() == "workspace missing or too small"
    assert b"gfx950" in L.lib.cogs_version()


def test_header_is_plain_c(tmp_path):
    """include/cogs.h is the boundary a C caller binds: it must compile as C99 on its own, every entry point a
    file-scope declaration (a function declared inside a struct body is C++ only and hides the symbol from C)"""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    src = tmp_path / "t.c"
    src.write_text('#include "cogs.h"\nint main(void) { cogs_gemm_desc d; cogs_vit_weights w; (void)d; (void)w;\n'
                   '  cogs_status (*f)(cogs_handle, int) = cogs_vit_set_streams; (void)f; return (int)(sizeof(cogs_kv) == 0); }\n')
    r = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-fsyntax-only",
                        "-I", os.path.join(ROOT, "include"), str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_no_cpu_fallback():
    """CPU tensors must be rejected loudly, never computed on the host"""
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from cogstream_amd import _lib as L
    from cogstream_amd import ops
    with pytest.raises(L.CogsError):
        ops.layernorm(torch.zeros(4, 64), torch.ones(64), torch.zeros(64))
    with pytest.raises(L.CogsError):
        ops.gemm(torch.zeros(4, 64, dtype=torch.bfloat16), torch.zeros(8, 64, dtype=torch.bfloat16))


def test_product_does_not_import_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "cogstream_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f"{f} imports the oracle"


def test_resize_table_and_patchify():
    """size table of SURVEY.md section 8 (computed from image_processing_videollama3.py:93-131)"""
    from cogstream_amd import processing as pr
    assert pr.simple_batched_resize([(224, 224)], 8) == [(224, 224)]
    assert pr.simple_batched_resize([(480, 854)], 64) == [(308, 588)]
    assert pr.simple_batched_resize([(480, 854)], 256) == [(140, 280)]
    assert pr.simple_batched_resize([(480, 640)], 64) == [(364, 504)]
    # patchify round trip: row r <-> (frame, merge-row, merge-col, window, c, py, px)
    t, gh, gw, ms = 2, 4, 6, 2
    img = np.arange(t * 3 * gh * 14 * gw * 14, dtype=np.float32).reshape(t, 3, gh * 14, gw * 14)
    rows = pr.patchify(img, ms)
    assert rows.shape == (t * gh * gw, 588)
    from oracle import vision as ov
    order = ov.patchify_order(t, gh, gw, ms)
    for r in (0, 1, 2, 3, 4, 17, 47):
        f, pr_, pc = order[r].tolist()
        patch = img[f, :, pr_ * 14:(pr_ + 1) * 14, pc * 14:(pc + 1) * 14].reshape(-1)
        assert np.array_equal(rows[r], patch)


def test_preprocess_synthetic_clip_shapes():
    from cogstream_amd import processing as pr
    frames, ts = pr.synthetic_clip(8, 224, 224)
    out = pr.preprocess_videos([frames])
    assert out["pixel_values"].shape == (2048, 588) and out["grid_sizes"].tolist() == [[8, 16, 16]]
    assert out["pixel_values"].dtype == np.float32 and abs(out["pixel_values"]).max() <= 1.0
    d, _ = pr.synthetic_clip(4, 56, 56, kind="drift")
    assert d.shape == (4, 56, 56, 3)


def test_chat_template_rendering():
    """SURVEY.md appendix B3 (rendered by the reference's chat template)"""
    from cogstream_amd import processing as pr
    conv = [{"role": "system", "content": "You are a helpful assistant."},
            {"role": "user", "content": [{"type": "video", "num_frames": 3, "timestamps": [0.0, 1.0, 2.04]},
                                         {"type": "text", "text": "Q1?"}]},
            {"role": "assistant", "content": "A1."},
            {"role": "user", "content": "Q2?"}]
    want = ("<|im_start|>system\nYou are a helpful assistant.<|im_end|>\n<|im_start|>user\nTime 0.0s:<image>,Time 1.0s:<image>,"
            "Time 2.0s:<image>\nQ1?<|im_end|>\n<|im_start|>assistant\nA1.<|im_end|>\n<|im_start|>user\nQ2?<|im_end|>\n"
            "<|im_start|>assistant\n")
    assert pr.render_conversation(conv) == want
    assert pr.expand_image_tokens("a<image>b<image>", [2, 1]) == "a<image><image>b<image>"
    assert pr.process_history_qas(conv) == (["Q1?"], ["A1."], "Q2?")
    no_sys = pr.render_conversation(conv[1:2], add_generation_prompt=False)
    assert no_sys.startswith("<|im_start|>system\nYou are VideoLLaMA3 created by Alibaba DAMO Academy")


def test_processor_call_contract():
    from cogstream_amd import processing as pr
    from toy_tokenizer import IMAGE, ToyTokenizer
    frames, ts = pr.synthetic_clip(4, 56, 56)
    conv = [{"role": "user", "content": [{"type": "video", "video": frames, "timestamps": ts}, {"type": "text", "text": "What?"}]}]
    out = pr.CogStreamProcessor(ToyTokenizer())(conversation=conv, add_system_prompt=True, add_generation_prompt=True)
    for k in ("input_ids", "attention_mask", "pixel_values", "grid_sizes", "merge_sizes", "modals", "tokenizer", "hist_qs",
              "hist_as", "current_question", "all_timestamps", "total_image_num", "original_text"):
        assert k in out, k
    gs, ms = out["grid_sizes"], out["merge_sizes"]
    n_tok = int((gs.prod(1) // (ms ** 2)).sum())
    assert int((out["input_ids"] == IMAGE).sum()) == n_tok          # processing_cogreasoner.py:606,727
    assert out["total_image_num"] == 4 and out["current_question"] == "What?" and out["hist_qs"] == []


def test_frame_shards():
    from cogstream_amd.parallel import frame_shards
    assert frame_shards(64, 8) == [(8 * i, 8 * i + 8) for i in range(8)]
    assert frame_shards(10, 4) == [(0, 3), (3, 6), (6, 8), (8, 10)]


def test_pillow_resample_restatement_is_bit_exact():
    """cogstream_amd.processing.resize_bicubic_exact (the algorithm the GPU kernels implement) against PIL itself"""
    from PIL import Image
    from cogstream_amd import processing as pr
    rng = np.random.default_rng(1)
    for (h, w, th, tw) in [(120, 214, 84, 140), (60, 60, 60, 60), (33, 47, 56, 84), (96, 50, 28, 28)]:
        f = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        ref = np.asarray(Image.fromarray(f).resize((tw, th), resample=Image.BICUBIC))
        assert np.array_equal(pr.resize_bicubic_exact(f, (th, tw)), ref), (h, w, th, tw)


def test_result_json_layout(tmp_path):
    """the on-disk result format of evaluate/answer_generate.py:30-35,143 (what eval_metrics consumes)"""
    import json
    from cogstream_amd.answer_generate import save_to_json
    recs = [{"qa_id": 0, "question": "q", "answer": "a", "prediction": "p\u00e9", "predicted_coi": [], "predicted_visual": True, "coi": []}]
    path = save_to_json("clip_007", [recs], str(tmp_path / "out"))
    d = json.load(open(path, encoding="utf-8"))
    assert path.endswith("clip_007.json") and d == {"video_name": "clip_007", "Data": [recs]}
    text = open(path, encoding="utf-8").read()
    assert "\\u00e9" not in text and "\u00e9" in text               # ensure_ascii=False like the reference


def test_cfg4_history_fixture_and_retrieval_prompt():
    """the committed cfg4 session strings drive the retrieval prompt builder without surprises"""
    import json
    from cogstream_amd.qaselect import format_example
    turns = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "cfg4_history.json")))["turns"]
    assert len(turns) == 8 and all(t["question"] and t["answer"] for t in turns)
    prompt = format_example({"current_Q": turns[-1]["question"], "hist_Qs": [t["question"] for t in turns[:-1]],
                             "hist_As": [t["answer"] for t in turns[:-1]]})
    for i, t in enumerate(turns[:-1]):
        assert t["question"] in prompt and t["answer"] in prompt
    assert turns[-1]["question"] in prompt


def test_reference_plugin_names_resolve_through_auto_map(tmp_path):
    """The reference's only extension API is HF auto_map: model/config.json:6-9 names
    cogreasoner_chat.Videollama3Qwen2ForCausalLM, model/preprocessor_config.json:2-5 names
    processing_cogreasoner.Videollama3Qwen2Processor (used by evaluate/answer_generate.py:173-183). Both names exist in
    this package as modules + classes, and cogstream_amd.auto resolves a checkpoint directory's auto_map to them --
    checked on the reference's own JSON files where they exist, and on a directory written by save_checkpoint."""
    import json
    from cogstream_amd import auto
    from cogstream_amd.cogreasoner_chat import Videollama3Qwen2ForCausalLM
    from cogstream_amd.processing_cogreasoner import Videollama3Qwen2Processor
    from cogstream_amd.chat import CogReasoner
    from cogstream_amd.processing import CogStreamProcessor
    assert issubclass(Videollama3Qwen2ForCausalLM, CogReasoner) and issubclass(Videollama3Qwen2Processor, CogStreamProcessor)
    for name in ("qa_selection", "generate", "set_adapter", "load_adapter", "to", "eval", "from_pretrained"):
        assert callable(getattr(Videollama3Qwen2ForCausalLM, name))
    for name in ("__call__", "batch_decode", "from_pretrained"):
        assert callable(getattr(Videollama3Qwen2Processor, name))
    d = tmp_path / "ckpt"
    d.mkdir()
    json.dump({"auto_map": {"AutoModelForCausalLM": "cogreasoner_chat.Videollama3Qwen2ForCausalLM"}}, open(d / "config.json", "w"))
    json.dump({"auto_map": {"AutoProcessor": "processing_cogreasoner.Videollama3Qwen2Processor"}},
              open(d / "preprocessor_config.json", "w"))
    assert auto._resolve(str(d), "config.json", "AutoModelForCausalLM", "x.y") is Videollama3Qwen2ForCausalLM
    assert auto._resolve(str(d), "preprocessor_config.json", "AutoProcessor", "x.y") is Videollama3Qwen2Processor
    json.dump({"auto_map": {"AutoModelForCausalLM": "modeling_other.SomethingElse"}}, open(d / "config.json", "w"))
    with pytest.raises(ValueError):
        auto._resolve(str(d), "config.json", "AutoModelForCausalLM", "x.y")
    ref = "/root/reference/model"
    if os.path.isdir(ref):      # build container only: the reference's own files name exactly these classes
        assert auto._resolve(ref, "config.json", "AutoModelForCausalLM", "x.y") is Videollama3Qwen2ForCausalLM
        assert auto._resolve(ref, "preprocessor_config.json", "AutoProcessor", "x.y") is Videollama3Qwen2Processor
