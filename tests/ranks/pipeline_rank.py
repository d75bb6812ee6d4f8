"""One rank of the frame-sharded product path (BASELINE configs[2] in miniature), started by
tests/test_gpu_multirank.py under torch.distributed.run: the tiny model of the e2e fixtures, fp32, per-frame
(block-diagonal) attention -- the semantics the reference ships with on GPUs and the only one that shards.

    pipeline_rank.py <helper|spread> <projected|encoder>

helper: rank 0 holds the whole model, every other rank is built WITHOUT the Qwen2 weights (llm=None);
spread: every rank holds the model and the K + 1 event-summary sequences are spread too.
Rank 0 checks what the reference produced in that mode (tests/golden/e2e_blockdiag.npz, cases b and c: cluster
indices, minor frames, keep-mask, greedy tokens); every rank prints the tokens it returned. One-GPU rehearsal: all
ranks share cuda:0 and the collectives run over gloo (staged through the host)."""
import json
import os
import random
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.environ["COGS_ROOT"]
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from golden.inputs import FORCED_COSINE, e2e_inputs  # noqa: E402
from toy_tokenizer import ToyTokenizer  # noqa: E402

from cogstream_amd.chat import CogReasoner  # noqa: E402
from cogstream_amd.llm import Qwen2Engine  # noqa: E402
from cogstream_amd.vision import BLOCK_DIAG, Projector, VisionEncoder  # noqa: E402
from cogstream_amd.weights import LlmConfig, VisionConfig, random_llm_state, random_proj_state, random_vit_state  # noqa: E402

VIT = dict(hidden_size=576, intermediate_size=200, num_hidden_layers=2, num_attention_heads=8)
LLM = dict(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2,
           num_key_value_heads=1, vocab_size=512, image_token_index=258, eos_token_id=257)


def main():
    mode, payload = sys.argv[1], sys.argv[2]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")
    dev = torch.device("cuda:0")
    vcfg, lcfg = VisionConfig(**VIT), LlmConfig(**LLM)
    dt = torch.float32
    enc = VisionEncoder(random_vit_state(vcfg, seed=3, std=0.05), vcfg, dtype=dt, device=dev, attn_mode=BLOCK_DIAG)
    proj = Projector(random_proj_state(vcfg.hidden_size, lcfg.hidden_size, seed=1, std=0.05), dtype=dt, device=dev)
    with_llm = rank == 0 or mode == "spread"
    eng = Qwen2Engine(random_llm_state(lcfg, seed=7, std=0.05), lcfg, dtype=dt, device=dev) if with_llm else None
    model = CogReasoner(enc, proj, eng, lcfg, generation_config=dict(do_sample=False, eos_token_id=[257]))
    model.enable_sharded_encoder(rank, world, payload=payload)
    if mode == "spread":
        model.enable_distributed_events(rank, world)
    g = {k: v for k, v in np.load(os.path.join(ROOT, "tests", "golden", "e2e_blockdiag.npz")).items()}
    tok = ToyTokenizer()
    ok = True
    records = []
    for tag in ("b", "c"):
        inp = e2e_inputs(tag)
        ids = tok(inp["text"])
        # only rank 0's host generators matter (it runs k-means); the others are deliberately seeded differently
        random.seed(5 if rank == 0 else 1000 + rank)
        torch.manual_seed(5 if rank == 0 else 1000 + rank)
        sel = model.qa_selection(current_question=inp["current_question"], hist_qs=inp["hist_qs"], hist_as=inp["hist_as"],
                                 tokenizer=tok, original_text=inp["text"], input_ids=ids["input_ids"],
                                 attention_mask=ids["attention_mask"], mode="FCC", all_timestamps=inp["timestamps"])
        model.cosine_override = FORCED_COSINE if tag == "c" else None
        out, _ = model.generate(pixel_values=inp["pixel_values"], grid_sizes=inp["grid_sizes"], merge_sizes=inp["merge_sizes"],
                                modals=["video"], new_input_ids=sel["new_input_ids"], new_attention_mask=sel["new_attention_mask"],
                                selection_module_output=sel["selection_module_output"], if_visual=sel["if_visual"],
                                total_image_num=inp["T"], max_new_tokens=8, repetition_penalty=1.05)
        toks = out[0].tolist()
        rec = {"rank": rank, "tag": tag, "tokens": toks, "pieces": model.last_debug.get("shard_pieces")}
        if rank == 0:
            d = model.last_debug
            checks = {"tokens": toks == g[f"{tag}_tokens"].tolist(), "minor": d["minor_frames"] == g[f"{tag}_minor"].tolist(),
                      "mask": bool(torch.equal(d["compression_mask"].cpu(), torch.from_numpy(g[f"{tag}_mask"]))),
                      "assign": d["assign"] == g[f"{tag}_assign"].tolist(),
                      "cosine": float((d["cosine_raw"] - torch.from_numpy(g[f"{tag}_cosine"])).abs().max()) < 1e-3}
            rec["checks"] = checks
            ok = ok and all(checks.values())
        records.append(rec)
    every = [None] * world
    dist.all_gather_object(every, records)       # one writer: lines of two processes on one pipe can interleave
    if rank == 0:
        for recs in every:
            for rec in recs:
                print("RANK_RECORD " + json.dumps(rec), flush=True)
        print("PIPELINE_RESULT", "OK" if ok else "MISMATCH", flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
