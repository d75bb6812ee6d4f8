"""Multi-turn streaming driver: the caller contract of evaluate/answer_generate.py (infer :60-76, inference
:102-150) for in-memory clips. One process per GPU; with world_size > 1 the VIDEOS are sharded over the ranks
exactly like the reference's DistributedSampler (:186-187) -- independent replicas, no collective ("cfg5").

A session is a list of segments; every segment adds a clip (uint8 [t,H,W,3] + timestamps) and one or more
questions. Each question is answered with the whole conversation so far (all clips, selected history):
    processor(conversation) -> model.qa_selection(mode="FCC") -> model.generate() -> decode
and the answer is appended as an assistant turn, as the reference does."""
from __future__ import annotations

from typing import Any, Dict, List, Optional, Sequence

import torch

from .chat import parse_selection


@torch.inference_mode()
def infer(conversation, model, processor, select=None, if_visual=None, max_new_tokens: int = 1024, **gen_kwargs):
    """evaluate/answer_generate.py:60-76"""
    inputs = processor(conversation=conversation, add_system_prompt=True, add_generation_prompt=True, return_tensors="pt")
    inputs = {k: (v.to(model.device) if isinstance(v, torch.Tensor) and k in ("pixel_values",) else v) for k, v in inputs.items()}
    if "pixel_values" in inputs and model.dtype == torch.bfloat16:
        inputs["pixel_values"] = inputs["pixel_values"].to(dtype=torch.bfloat16)   # :70
    # :71-73 -- the selection stage runs under the "language_module" adapter, the answer under "full_module";
    # a model without adapters (none have been released) runs both on the base weights
    has = getattr(model, "_adapters", {})
    if "language_module" in has:
        model.set_adapter("language_module")
    inputs = model.qa_selection(**inputs, mode="FCC", select_gt=select, if_visual=if_visual)
    if "full_module" in has:
        model.set_adapter("full_module")
    output_ids, selection = model.generate(**inputs, max_new_tokens=max_new_tokens, **gen_kwargs)
    response = processor.batch_decode(output_ids, skip_special_tokens=True)[0].strip()
    return response, selection


def run_session(model, processor, segments: Sequence[Dict[str, Any]], system: str = "You are a helpful assistant.",
                max_new_tokens: int = 1024, **gen_kwargs) -> List[Dict[str, Any]]:
    """segments: [{"video": uint8 [t,H,W,3], "timestamps": [...], "questions": [str, ...],
                   "answers": [str, ...] (optional ground truth), "relevance": [...] (optional)}]
    -> per-question records with the reference's result keys (:143)."""
    conversation: List[Dict[str, Any]] = [{"role": "system", "content": system}]
    records: List[Dict[str, Any]] = []
    hist = 0
    for seg in segments:
        for i, q in enumerate(seg["questions"]):
            if i == 0:
                conversation.append({"role": "user", "content": [
                    {"type": "video", "video": seg["video"], "timestamps": list(seg["timestamps"])},
                    {"type": "text", "text": q}]})
            else:
                conversation.append({"role": "user", "content": q})
            out, selection = infer(conversation, model, processor, max_new_tokens=max_new_tokens, **gen_kwargs)
            if hist > 0:
                vis, idx = parse_selection(selection)
                relevance = [1 if j in idx else 0 for j in range(hist)]
            else:
                vis, relevance = True, []
            gt = seg.get("answers", [None] * len(seg["questions"]))[i]
            records.append({"qa_id": hist, "question": q, "answer": gt, "prediction": out, "predicted_coi": relevance,
                            "predicted_visual": vis, "coi": (seg.get("relevance") or [None] * len(seg["questions"]))[i]})
            hist += 1
            conversation.append({"role": "assistant", "content": out})
    return records


def save_to_json(video_name: str, data, folder_path: str) -> str:
    """evaluate/answer_generate.py:30-35 -- the result file the reference's eval_metrics reads:
    {"video_name": ..., "Data": [[record, ...]]} with one inner list per query chain"""
    import json
    import os
    os.makedirs(folder_path, exist_ok=True)
    file_path = os.path.join(folder_path, f"{video_name}.json")
    with open(file_path, "w", encoding="utf-8") as f:
        json.dump({"video_name": video_name, "Data": data}, f, ensure_ascii=False, indent=4)
    return file_path


def shard_videos(n_videos: int, rank: int, world: int) -> List[int]:
    """DistributedSampler(shuffle=False)-style round robin with padding by wrap-around (:186)"""
    per = -(-n_videos // world)
    idx = list(range(n_videos)) + list(range(per * world - n_videos))
    return idx[rank::world][:per]
