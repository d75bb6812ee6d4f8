"""Host text logic against the REAL Qwen2 tokenizer and the reference code driven with it, through the committed
record tests/golden/tokenizer.json (made in the build container by tests/golden/make_golden.py::golden_tokenizer;
the vocabulary is reference data and does not ship). A live variant runs where /root/reference exists."""
import json
import os
import types

import pytest
import torch

from replay_tokenizer import ReplayTokenizer, unrle

REF_MODEL = "/root/reference/model"


@pytest.fixture(scope="module")
def tok():
    return ReplayTokenizer()


def test_allowed_token_ids_are_the_reference_processors(tok):
    """StructuredLogitsProcessor (model/qaselect_module_predict.py:86-103); SURVEY.md A17 lists the same set"""
    from cogstream_amd import qaselect
    want = [11] + list(range(15, 25)) + [58, 60, 2152, 9693, 151645]
    assert tok.data["allowed_ids"] == want
    assert qaselect.allowed_token_ids(tok) == want
    assert tok.data["select_kwargs"] == {"max_new_tokens": qaselect.SELECT_MAX_NEW_TOKENS, "num_beams": 1,
                                         "do_sample": False, "eos_token_id": qaselect.SELECT_EOS}
    assert qaselect.normalise_selection("[yes,0]") == tok.data["select_roundtrip"]


def test_special_ids_and_appendix_b3_strings(tok):
    from cogstream_amd.chat import DEFAULT_GENERATION
    from cogstream_amd.weights import LlmConfig
    d = tok.data
    assert d["image_token_id"] == LlmConfig().image_token_index == 151665
    assert d["im_end"] == LlmConfig().eos_token_id == 151645 and d["endoftext"] == 151643
    assert DEFAULT_GENERATION["eos_token_id"] == [d["im_end"], d["endoftext"]] and DEFAULT_GENERATION["pad_token_id"] == d["endoftext"]
    s = d["strings"]
    assert s["Time 12.0s:"] == [1462, 220, 16, 17, 13, 15, 82, 25] and s[","] == [11] and s["\n"] == [198]
    assert s["<|im_start|>assistant\n"] == [151644, 77091, 198]


def test_chat_template_equals_the_references_jinja_rendering(tok):
    """model/chat_template.json rendered by the tokenizer's jinja engine, 3 conversations x (system prompt, generation
    prompt) switches, against cogstream_amd.processing.render_conversation"""
    from cogstream_amd.processing import render_conversation
    assert len(tok.data["templates"]) == 12
    for t in tok.data["templates"]:
        got = render_conversation(t["conversation"], t["add_system_prompt"], t["add_generation_prompt"])
        assert got == t["text"], (t["name"], t["add_system_prompt"], t["add_generation_prompt"])


def test_prompt_lengths_and_the_bench_formula(tok):
    """cfg2 / cfg1 / cfg3 prompts tokenised by the real tokenizer: S = 30 (default system turn) + 3 (user header)
    + sum over frames of (len('Time t.0s:') + P + 1) + question (7) + <|im_end|>\\n (2) + generation prompt (3)"""
    from bench import prompt_tokens
    from cogstream_amd.processing import expand_image_tokens, render_conversation
    for p in tok.data["prompts"]:
        T, P = p["T"], p["P"]
        conv = [{"role": "user", "content": [{"type": "video", "num_frames": T, "timestamps": [float(i) for i in range(T)]},
                                             {"type": "text", "text": p["question"]}]}]
        text = expand_image_tokens(render_conversation(conv), [P] * T)
        ids = tok.encode(text)                                   # KeyError if the text differs from the reference's
        assert len(ids) == p["len"] and ids == unrle(p["ids_rle"])
        assert sum(1 for i in ids if i == 151665) == T * P == p["n_image"]
        assert prompt_tokens(T, P) == p["len"], (T, P)
    assert [p["len"] for p in tok.data["prompts"]] == [15395, 621, 15295]


def test_prompt_surgery_with_real_token_ids(tok):
    """prepare_inputs / process_input_ids (model/cogreasoner_chat.py:121-177,478-511) run by the reference with the
    real tokenizer; the product must build the same text (looked up) and so the same ids"""
    from cogstream_amd.chat import CogReasoner
    s = tok.data["surgery"]
    me = types.SimpleNamespace(hist_qs=s["hist_qs"], hist_as=s["hist_as"], current_question=s["current_question"], tokenizer=tok)
    for c in s["cases"]:
        enc, if_visual = CogReasoner.prepare_inputs(me, c["selection"], original_text=s["original_text"])
        assert if_visual == c["if_visual"]
        assert enc["input_ids"][0].tolist() == unrle(c["ids_rle"]), c["selection"]
        assert tok.decode(enc["input_ids"][0]) == c["prompt"]


def test_retrieval_and_summary_prompts_tokenise_like_the_references(tok):
    from cogstream_amd.chat import create_visual_summary_prompt
    from cogstream_amd.qaselect import format_example
    turns = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "cfg4_history.json")))["turns"]
    qs, as_ = [t["question"] for t in turns], [t["answer"] for t in turns]
    lens = [len(tok.encode(format_example({"current_Q": qs[n - 1], "hist_Qs": qs[:n - 1], "hist_As": as_[:n - 1]})))
            for n in range(1, 9)]
    assert lens == tok.data["cfg4_retrieval_prompt_lens"] == [462, 488, 510, 536, 562, 581, 610, 636]
    sp = create_visual_summary_prompt(15 * 50, torch.arange(15, dtype=torch.float32) + 30)
    assert len(tok.encode(sp)) == tok.data["summary_prompt_len"]


@pytest.mark.skipif(not os.path.isdir(REF_MODEL), reason="the checkpoint's tokenizer data only exists in the build container")
def test_live_real_tokenizer_agrees_with_the_record():
    """build container only: the replayed record IS what the real tokenizer produces today, and the product's
    processor runs end to end on the real tokenizer (8 x 224 x 224, BASELINE configs[0] plumbing)"""
    import numpy as np
    from transformers import Qwen2Tokenizer
    from cogstream_amd import processing as pr
    from cogstream_amd import qaselect
    real, rec = Qwen2Tokenizer.from_pretrained(REF_MODEL), ReplayTokenizer()
    for text, ids in list(rec.by_text.items())[:40]:
        assert real.encode(text, add_special_tokens=False) == ids
    assert qaselect.allowed_token_ids(real) == rec.data["allowed_ids"]
    frames, ts = pr.synthetic_clip(8, 224, 224)
    conv = [{"role": "user", "content": [{"type": "video", "video": frames, "timestamps": ts},
                                         {"type": "text", "text": "What is happening in the video?"}]}]
    out = pr.CogStreamProcessor(real)(conversation=conv, add_system_prompt=True, add_generation_prompt=True)
    assert out["input_ids"].shape[1] == rec.data["prompts"][1]["len"] == 621
    assert out["input_ids"][0].tolist() == unrle(rec.data["prompts"][1]["ids_rle"])
