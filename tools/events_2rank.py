#!/usr/bin/env python3
"""One-GPU rehearsal of the sharded event-summary passes (parallel.pooled_means_sharded): two gloo ranks on cuda:0 run
select_events_based_on_summary on a 256-frame clip's projected tokens; both must return the single-process result.
    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/events_2rank.py"""
import os
import random
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from cogstream_amd.chat import CogReasoner  # noqa: E402
from cogstream_amd.llm import Qwen2Engine  # noqa: E402
from cogstream_amd.weights import LlmConfig, random_llm_state  # noqa: E402
from toy_tokenizer import IM_END, IMAGE, ToyTokenizer  # noqa: E402

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
dev = torch.device("cuda:0")
bf = torch.bfloat16
lcfg = LlmConfig(image_token_index=IMAGE, eos_token_id=IM_END, num_hidden_layers=int(os.environ.get("LAYERS", "28")))
eng = Qwen2Engine(random_llm_state(lcfg, 2, dev, bf), lcfg, dtype=bf, device=dev)
T, P = 256, 50
g = torch.Generator(device="cpu").manual_seed(4)
base = torch.randn(18, 1, 3584, generator=g)
mm = (base[torch.arange(T) * 18 // T] + 0.3 * torch.randn(T, P, 3584, generator=g)).reshape(T * P, 3584).to(dev, bf)
ts = torch.arange(T, dtype=torch.float32)


def run(distributed):
    model = CogReasoner(None, None, eng, lcfg, generation_config={})
    model.tokenizer, model.current_question = ToyTokenizer(), "What is happening in the video?"
    if distributed:
        model.enable_distributed_events(rank, world)
    random.seed(0)
    torch.manual_seed(0)
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    out = model.select_events_based_on_summary(mm, T, ts)
    torch.cuda.synchronize()
    dist.barrier()
    return out, time.perf_counter() - t0, model.last_debug["cosine_raw"]


run(False)
one, t_one, cos_one = run(False)
two, t_two, cos_two = run(True)
same = one == two and torch.equal(cos_one, cos_two)
flags = [None] * world
dist.all_gather_object(flags, same)
if rank == 0:
    print(f"events over {world} ranks (both on one GPU, gloo): identical on every rank: {all(flags)}; "
          f"{len(one)} minor frames; single {t_one * 1e3:.0f} ms, sharded {t_two * 1e3:.0f} ms (same GPU: no speed-up expected)")
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if all(flags) else 1)
