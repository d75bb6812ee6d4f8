"""Historic Dialogue Retrieval: prompt + constrained greedy decode (model/qaselect_module_predict.py:5-127).

The prompt text and the allowed-token set are part of the model's behaviour (the selection adapter was
trained on exactly this prompt), so they are reproduced verbatim as data; the decode itself runs on the
HIP Qwen2 engine with the allowed-id logits mask applied on the device (cogs_logits_process)."""
from __future__ import annotations

from typing import List, Sequence

_SYSTEM = (
    "<|im_start|>system\n"
    "You are a QA-pair filtering assistant. Your task is to identify which of the historical QA pairs are helpful "
    "for answering the current question and determine if the historical QA pairs alone are sufficient to answer it.\n"
    "\n"
    "A QA pair is considered helpful if it provides:\n"
    "- Relevant background information, context, or details\n"
    "- Additional facts or insights that can be used to answer the current question\n"
    "- Matching roles, scenarios, or domain knowledge that could support the answer\n"
    "\n"
    "Output a single bracketed sequence:\n"
    "- Start with 'yes' if the historical QA pairs are insufficient to fully answer the question (additional visual "
    "information may be needed).\n"
    "- Start with 'no' if the current question can be fully answered using only the historical QA pairs (no additional "
    "visual information needed).\n"
    "- Follow with the indices (starting from 0) of the helpful QA pairs, e.g., [yes,0,5] or [no,0,5].\n"
    "- If no QA pairs are helpful, output [yes] or [no] based on the question's dependency.\n"
    "- Do not add extra text or explanation — only output the bracketed sequence.\n"
    "<|im_end|>"
)

_HIST = ("0. Q: How to measure earthquakes? A: Using the Richter scale\n"
         "1. Q: What is tectonic plate? A: Massive rock slabs beneath crust\n"
         "2. Q: What is the weather like today? A: Sunny and warm\n")

_DEMO = (
    "\nExample:\n"
    "Current Question: What causes earthquakes?\n"
    "Historical QA Pairs:\n" + _HIST +
    "→ Output: [no,1]\n"
    "------------------------------\n"
    "Example:\n"
    "Current Question: What does an earthquake look like?\n"
    "Historical QA Pairs:\n" + _HIST +
    "→ Output: [yes]\n"
    "------------------------------"
)

_TAIL = ("\nGenerate a bracketed sequence (e.g., [yes,0,5] or [no,0,5]) indicating the dependency (yes or no) and the "
         "indices of helpful QA pairs. Only output the bracketed sequence.")

ALLOWED_STRINGS = [str(i) for i in range(10)] + ["[", "]", ",", "<|im_end|>", "no", "yes"]
SELECT_EOS = 151645
SELECT_MAX_NEW_TOKENS = 50


def format_example(example: dict, include_demo: bool = True) -> str:
    """qaselect_module_predict.py:5-61"""
    user = (_DEMO if include_demo else "") + f"\nCurrent Question: {example['current_Q']}\n\nHistorical QA Pairs (ordered by time):"
    for i, (q, a) in enumerate(zip(example["hist_Qs"], example["hist_As"])):
        user += f"\n{i}. Q: {q}\n   A: {a}"
    user += _TAIL
    return f"{_SYSTEM}<|im_start|>user\n{user}<|im_end|><|im_start|>assistant\n"


def allowed_token_ids(tokenizer) -> List[int]:
    """StructuredLogitsProcessor._get_allowed_token_ids (:90-98); {11, 15-24, 58, 60, 2152, 9693, 151645}
    with the Qwen2 tokenizer"""
    ids = set()
    for t in ALLOWED_STRINGS:
        for i in tokenizer.encode(t, add_special_tokens=False):
            if i >= 0:
                ids.add(int(i))
    return sorted(ids)


def normalise_selection(text: str) -> str:
    """:121-126"""
    text = text.strip()
    if text == "":
        text = "[yes]"
    if not text.endswith("]"):
        text += "]"
    if not text.startswith("["):
        text = "[" + text
    return text


def select_qas(current_question: str, hist_Qs: Sequence[str], hist_As: Sequence[str], model, tokenizer=None,
               include_demo: bool = True) -> str:
    """:63-127; `model` is a cogstream_amd.chat.CogReasoner (needs .generate_language_module)"""
    if tokenizer is None:
        raise ValueError("If passing a model instance, please provide a tokenizer as well.")
    prompt = format_example({"current_Q": current_question, "hist_Qs": list(hist_Qs), "hist_As": list(hist_As)},
                            include_demo=include_demo)
    enc = tokenizer(prompt, return_tensors="pt")
    ids = enc["input_ids"]
    new = model.generate_language_module(input_ids=ids, attention_mask=enc.get("attention_mask"),
                                         max_new_tokens=SELECT_MAX_NEW_TOKENS, do_sample=False,
                                         allowed_ids=allowed_token_ids(tokenizer), eos_token_id=SELECT_EOS)
    gen = new[0, ids.shape[1]:]
    return normalise_selection(tokenizer.decode(gen, skip_special_tokens=True))
