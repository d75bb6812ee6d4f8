"""Multi-turn streaming driver: the caller contract of evaluate/answer_generate.py (infer :60-76, inference
:102-150, __main__ :160-190). One process per GPU, started by the reference's own launcher line

    torchrun --nproc_per_node=8 -m cogstream_amd.answer_generate --model_path M --video_dir V --query_dir Q --save_dir S
             [--lora_adapter_1_path A1 --lora_adapter_2_path A2] [--mode replicas|shard]

(RANK / WORLD_SIZE / LOCAL_RANK from the environment, :169-171; a plain `python -m ...` run is world size 1).

  --mode replicas  BASELINE configs[4], the reference's own mode: the VIDEOS are dealt over the ranks by
                   DistributedSampler(dataset, num_replicas, rank) (:186 -- shuffle=True, seed 0, padded by wrap-around),
                   every rank is an independent replica with the whole model, no collective; each rank writes the result
                   files of its videos.
  --mode shard     BASELINE configs[2] (build-only; the reference never shards a clip): every rank walks ALL videos in
                   step, the frames of each request are encoded data-parallel (CogReasoner.enable_sharded_encoder: one
                   all-gather per request), the LLM stages run on rank 0, which writes the result files; the other ranks
                   load no Qwen2 weights unless --spread_events also spreads the event-summary passes.

A session is a list of segments; every segment adds a clip and one or more questions. Each question is answered with
the whole conversation so far (all clips, selected history):
    processor(conversation) -> model.qa_selection(mode="FCC") -> model.generate() -> decode
and the answer is appended as an assistant turn, as the reference does. Clips are in-memory uint8 [t,H,W,3] arrays
(run_session) or decoded-frame files `segment_<n>.npz` in the video's directory (inference; video_io.py -- this image
has no ffmpeg, so containers are not decoded here)."""
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


def sampler_indices(n_videos: int, rank: int, world: int, shuffle: bool = True, seed: int = 0) -> List[int]:
    """torch.utils.data.DistributedSampler(dataset, num_replicas=world, rank=rank).__iter__ at epoch 0 (:186-187):
    randperm under manual_seed(seed) when shuffling, padded to a multiple of `world` by wrap-around, then rank::world"""
    if shuffle:
        g = torch.Generator()
        g.manual_seed(seed)
        idx = torch.randperm(n_videos, generator=g).tolist()
    else:
        idx = list(range(n_videos))
    total = -(-n_videos // world) * world
    pad = total - len(idx)
    if pad > 0:
        idx += (idx * (-(-pad // max(len(idx), 1))))[:pad]
    return idx[rank:total:world]


def natural_sort_segments(folder_path: str) -> List[str]:
    """:19-28 -- files of a video directory ordered by the number in `segment_<n>`"""
    import os
    import re
    pat = re.compile(r"segment_(\d+)")

    def key(name):
        m = pat.search(name)
        return int(m.group(1)) if m else 999999

    return sorted(os.listdir(folder_path), key=key)


class VideoDataset:
    """:78-100 -- one item per query file `<video_name>.json` whose video directory exists; the first query chain"""

    def __init__(self, video_dir: str, query_dir: str):
        import json
        import os
        self.items: List[Dict[str, Any]] = []
        for json_file in os.listdir(query_dir):          # listing order, like the reference
            if not json_file.endswith(".json"):
                continue
            video_path = os.path.join(video_dir, json_file[:-len(".json")])
            if os.path.exists(video_path):
                with open(os.path.join(query_dir, json_file), encoding="utf-8") as f:
                    self.items.append({"video_path": video_path, "query_chains": json.load(f)})
            else:
                print(f"warning: video {video_path} not found (query file {json_file})")
        print(f"{len(self.items)} samples in total.")

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        it = self.items[i]
        return {"video_path": it["video_path"], "query_chain": it["query_chains"][0]}


def _scalar(x):
    """query files hold plain values; the reference sees them wrapped by the DataLoader's collate (qa["Q"][0], .item())"""
    if isinstance(x, (list, tuple)) and len(x) == 1:
        return x[0]
    return x.item() if hasattr(x, "item") else x


def answer_video(model, processor, video_path: str, query_chain, max_new_tokens: int = 1024, **gen_kwargs):
    """the body of inference()'s loop for one video (:106-148) -> [[record, ...]] (`Data` of the result file)"""
    import os
    conversation: List[Dict[str, Any]] = [{"role": "system", "content": "You are a helpful assistant."}]
    segments: Dict[Any, list] = {}
    for qa in query_chain:
        segments.setdefault(_scalar(qa["info"]["Event_Time"]), []).append(qa)
    video_answer: List[Dict[str, Any]] = []
    hist = 0
    for t, file_name in zip(sorted(segments), natural_sort_segments(video_path)):
        qas = segments[t]
        cov = {"role": "user", "content": [
            {"type": "video", "video": {"video_path": os.path.join(video_path, file_name), "fps": 1, "max_frames": 180}},
            {"type": "text", "text": _scalar(qas[0]["Q"])}]}
        for i, qa in enumerate(qas):
            conversation.append(cov if i == 0 else {"role": "user", "content": _scalar(qa["Q"])})
            output, selection = infer(conversation, model, processor, max_new_tokens=max_new_tokens, **gen_kwargs)
            if hist > 0:
                vis, idx = parse_selection(selection)
                relevance = [1 if j in idx else 0 for j in range(hist)]
            else:
                vis, relevance = True, []
            video_answer.append({"qa_id": hist, "question": _scalar(qa["Q"]), "answer": _scalar(qa["A"]), "prediction": output,
                                 "predicted_coi": relevance, "predicted_visual": vis,
                                 "coi": qa["info"]["relevance"]})      # the file's own value (the reference's [0] undoes the collate)
            hist += 1
            conversation.append({"role": "assistant", "content": output})
    return [video_answer]


def inference(model, dataset: VideoDataset, indices: Sequence[int], processor, save_dir: Optional[str],
              max_new_tokens: int = 1024, **gen_kwargs) -> List[str]:
    """:102-150 over this rank's videos; save_dir=None: answer but write nothing (non-writing ranks of a sharded run)"""
    import os
    written = []
    for i in indices:
        item = dataset[i]
        data = answer_video(model, processor, item["video_path"], item["query_chain"], max_new_tokens, **gen_kwargs)
        if save_dir is not None:
            written.append(save_to_json(os.path.basename(item["video_path"]), data, save_dir))
    return written


def setup(rank: int, world_size: int, local_rank: int, backend: Optional[str] = None) -> None:
    """:154-158. backend: default nccl (= RCCL); COGS_DIST_BACKEND=gloo + COGS_ONE_GPU=1 is the one-GPU rehearsal"""
    import os
    import random
    import torch.distributed as dist
    backend = backend or os.environ.get("COGS_DIST_BACKEND", "nccl")
    one_gpu = os.environ.get("COGS_ONE_GPU") == "1"
    torch.cuda.set_device(0 if one_gpu else local_rank)
    if world_size > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world_size,
                                    device_id=torch.device("cuda", torch.cuda.current_device()))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world_size)
    torch.manual_seed(42 + rank)
    random.seed(42 + rank)


def main(argv=None) -> int:
    import argparse
    import os
    from .auto import AutoModelForCausalLM, AutoProcessor, PeftModel
    ap = argparse.ArgumentParser(description="CogReasoner inference (cogstream_amd), the reference driver's arguments.")
    ap.add_argument("--model_path", type=str, default="model", help="Path to the base model directory.")
    ap.add_argument("--lora_adapter_1_path", type=str, help="Path to the first LoRA adapter (loaded as full_module).")
    ap.add_argument("--lora_adapter_2_path", type=str, help="Path to the second LoRA adapter (loaded as language_module).")
    ap.add_argument("--video_dir", type=str, required=True, help="Directory containing the videos' segment folders.")
    ap.add_argument("--query_dir", type=str, required=True, help="Directory containing test query (QA) files.")
    ap.add_argument("--save_dir", type=str, default="evaluate/results", help="Directory to save the result.")
    ap.add_argument("--mode", choices=["replicas", "shard"], default="replicas")
    ap.add_argument("--payload", choices=["projected", "encoder"], default="projected", help="--mode shard: all-gather width")
    ap.add_argument("--spread_events", action="store_true", help="--mode shard: spread the event-summary passes too")
    ap.add_argument("--max_new_tokens", type=int, default=1024)
    ap.add_argument("--greedy", action="store_true", help="do_sample=False instead of the checkpoint's generation config")
    args = ap.parse_args(argv)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    setup(rank, world, local_rank)
    dev = torch.device("cuda", torch.cuda.current_device())
    shard = args.mode == "shard" and world > 1
    model = AutoModelForCausalLM.from_pretrained(args.model_path, trust_remote_code=True, torch_dtype=torch.bfloat16,
                                                 attn_implementation="flash_attention_2", device=dev,
                                                 load_llm=not shard or rank == 0 or args.spread_events)
    processor = AutoProcessor.from_pretrained(args.model_path, trust_remote_code=True, device=dev)
    if model.llm is not None:
        if args.lora_adapter_1_path:
            model = PeftModel.from_pretrained(model, args.lora_adapter_1_path, adapter_name="full_module")
        if args.lora_adapter_2_path:
            model.load_adapter(args.lora_adapter_2_path, adapter_name="language_module")
    model.to(dev)
    dataset = VideoDataset(args.video_dir, args.query_dir)
    gen = {"do_sample": False} if args.greedy else {}
    if shard:
        model.enable_sharded_encoder(rank, world, payload=args.payload)
        if args.spread_events:
            model.enable_distributed_events(rank, world)
        indices, save_dir = list(range(len(dataset))), (args.save_dir if rank == 0 else None)
    else:
        indices, save_dir = sampler_indices(len(dataset), rank, world), args.save_dir
    written = inference(model, dataset, indices, processor, save_dir, args.max_new_tokens, **gen)
    import sys
    sys.stdout.write(f"rank {rank}/{world} ({args.mode}): {len(indices)} videos answered, {len(written)} result files written\n")
    sys.stdout.flush()          # one write per line: the ranks share the launcher's pipe
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
