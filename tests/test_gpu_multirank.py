"""N > 1 through the PRODUCT path (BASELINE configs[2] and configs[4]) on the one GPU of the
test box: several processes started by `torch.distributed.run` share cuda:0, collectives over gloo (staged through the
host -- on a node the backend is nccl = RCCL and nothing changes above `torch.distributed`). Every test starts its ranks
as a plain command, the way a user / the driver would.

  * CogReasoner.enable_sharded_encoder: qa_selection -> generate on 2 ranks gives, on rank 0, the cluster indices, minor
    frames, keep-mask and greedy tokens the REFERENCE produced in per-frame-attention mode (e2e_blockdiag.npz) -- with
    encoder-only helpers and with the event passes spread, for both all-gather payloads; every rank returns the answer
  * python -m cogstream_amd.answer_generate: --mode replicas (configs[4]) and --mode shard (configs[2]) on 2 ranks write
    the same result files as one process
  * python bench.py --gpus 2 as a plain command starts its own ranks"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ENV = dict(os.environ, COGS_ROOT=ROOT, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT)


def _torchrun(nproc, port, *cmd, env=None, timeout=900):
    return subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
                           "--master-addr", "127.0.0.1", "--master-port", str(port), *cmd],
                          capture_output=True, text=True, timeout=timeout, env=env or ENV, cwd=ROOT)


def test_single_process_block_diagonal_pipeline_vs_reference(dev):
    """the production attention mode (per-frame) end to end against the reference run in that mode: cases b
    (k-means + 10 event passes) and c (forced cosines -> event compression) of e2e_blockdiag.npz, fp32 parity mode"""
    import random
    from golden.inputs import FORCED_COSINE, e2e_inputs
    from test_gpu_golden import _tiny_model
    from toy_tokenizer import ToyTokenizer
    g = {k: v for k, v in np.load(os.path.join(ROOT, "tests", "golden", "e2e_blockdiag.npz")).items()}
    model = _tiny_model(dev, torch.float32, 0)
    tok = ToyTokenizer()
    for tag in ("b", "c"):
        inp = e2e_inputs(tag)
        assert abs(float(inp["pixel_values"].double().abs().sum()) - float(g[f"{tag}_pix_checksum"])) < 1e-6
        ids = tok(inp["text"])
        random.seed(5)
        torch.manual_seed(5)
        sel = model.qa_selection(current_question=inp["current_question"], hist_qs=[], hist_as=[], tokenizer=tok,
                                 original_text=inp["text"], input_ids=ids["input_ids"], attention_mask=ids["attention_mask"],
                                 mode="FCC", all_timestamps=inp["timestamps"])
        model.cosine_override = FORCED_COSINE if tag == "c" else None
        out, _ = model.generate(pixel_values=inp["pixel_values"], grid_sizes=inp["grid_sizes"], merge_sizes=inp["merge_sizes"],
                                modals=["video"], new_input_ids=sel["new_input_ids"], new_attention_mask=sel["new_attention_mask"],
                                if_visual=True, total_image_num=inp["T"], max_new_tokens=8, repetition_penalty=1.05)
        d = model.last_debug
        assert d["assign"] == g[f"{tag}_assign"].tolist()
        assert d["minor_frames"] == g[f"{tag}_minor"].tolist()
        assert torch.equal(d["compression_mask"].cpu(), torch.from_numpy(g[f"{tag}_mask"]))
        assert float((d["cosine_raw"] - torch.from_numpy(g[f"{tag}_cosine"])).abs().max()) < 1e-3
        assert out[0].tolist() == g[f"{tag}_tokens"].tolist()


@pytest.mark.parametrize("mode,payload,port", [("helper", "projected", 29531), ("spread", "encoder", 29532)])
def test_two_rank_sharded_pipeline_vs_reference(dev, mode, payload, port):
    r = _torchrun(2, port, os.path.join(ROOT, "tests", "ranks", "pipeline_rank.py"), mode, payload)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "PIPELINE_RESULT OK" in r.stdout, r.stdout[-3000:]
    recs = [json.loads(ln.split(" ", 1)[1]) for ln in r.stdout.splitlines() if ln.startswith("RANK_RECORD ")]
    assert len(recs) == 4
    for tag in ("b", "c"):
        a, b = [x for x in recs if x["tag"] == tag and x["rank"] == 0][0], [x for x in recs if x["tag"] == tag and x["rank"] == 1][0]
        assert a["tokens"] == b["tokens"] and len(a["tokens"]) == 8          # the answer reaches every rank
        assert a["pieces"] == [[0, 0, 75]] and b["pieces"] == [[0, 75, 150]]  # 150 frames, 75 per rank


def _make_dataset(tmp_path, n_videos=4):
    """a synthetic checkpoint directory (tiny widths, byte-level Qwen2 tokenizer files) + `n_videos` videos of two
    decoded segments each (6 frames at 2 fps -> 3 frames at the driver's 1 fps) with two / one questions"""
    from cogstream_amd import checkpoint as ck
    from cogstream_amd import processing as pr
    from cogstream_amd.video_io import write_decoded_video
    from cogstream_amd.weights import LlmConfig, VisionConfig, random_llm_state, random_proj_state, random_vit_state
    vcfg = VisionConfig(hidden_size=576, intermediate_size=200, num_hidden_layers=2, num_attention_heads=8)
    lcfg = LlmConfig(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=1,
                     vocab_size=512, image_token_index=258, eos_token_id=257)
    model_path = str(tmp_path / "ckpt")
    ck.save_checkpoint(model_path, random_vit_state(vcfg, seed=3, std=0.05),
                       random_proj_state(vcfg.hidden_size, lcfg.hidden_size, seed=1, std=0.05),
                       random_llm_state(lcfg, seed=7, std=0.05), vcfg, lcfg,
                       generation={"do_sample": False, "eos_token_id": [257]}, n_shards=2)
    ck.save_byte_tokenizer(model_path)
    vdir, qdir = tmp_path / "videos", tmp_path / "queries"
    qdir.mkdir()
    for v in range(n_videos):
        d = vdir / f"clip_{v}"
        d.mkdir(parents=True)
        for s in range(2):
            fr, _ = pr.synthetic_clip(6, 56, 84, kind="drift", clip_idx=10 * v + s)
            write_decoded_video(str(d / f"segment_{s}.npz"), fr, native_fps=2.0)
        chain = [{"Q": f"What happens first in clip {v}?", "A": "a", "info": {"Event_Time": 3, "relevance": []}},
                 {"Q": "And what is on the left?", "A": "b", "info": {"Event_Time": 3, "relevance": [1]}},
                 {"Q": "What changes afterwards?", "A": "c", "info": {"Event_Time": 6, "relevance": [0, 1]}}]
        json.dump([chain], open(qdir / f"clip_{v}.json", "w"))
    return model_path, str(vdir), str(qdir)


def test_answer_generate_driver_replicas_and_shard_modes(dev, tmp_path):
    """BASELINE configs[4] (one video per GPU, the reference's DistributedSampler mode) and configs[2] (frames of every
    request sharded) through `python -m cogstream_amd.answer_generate` with the reference driver's arguments: 4 videos
    x 3 questions on 2 ranks; both modes write the 4 result files one process writes, record for record"""
    model_path, vdir, qdir = _make_dataset(tmp_path)
    common = ["--model_path", model_path, "--video_dir", vdir, "--query_dir", qdir, "--max_new_tokens", "5", "--greedy"]
    env = dict(ENV, COGS_DIST_BACKEND="gloo", COGS_ONE_GPU="1")
    outs = {}
    r = subprocess.run([sys.executable, "-m", "cogstream_amd.answer_generate", *common, "--save_dir", str(tmp_path / "one")],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    for name, port, extra in (("replicas", 29541, ["--mode", "replicas"]), ("shard", 29542, ["--mode", "shard"]),
                              ("shard_spread", 29543, ["--mode", "shard", "--spread_events", "--payload", "encoder"])):
        r = _torchrun(2, port, "-m", "cogstream_amd.answer_generate", *common, *extra, "--save_dir", str(tmp_path / name), env=env)
        assert r.returncode == 0, (name, r.stderr[-3000:])
        outs[name] = r.stdout
    assert "rank 1/2 (replicas): 2 videos answered, 2 result files written" in outs["replicas"]
    assert "rank 1/2 (shard): 4 videos answered, 0 result files written" in outs["shard"]
    ref = {}
    for v in range(4):
        d = json.load(open(tmp_path / "one" / f"clip_{v}.json"))
        assert d["video_name"] == f"clip_{v}" and len(d["Data"]) == 1 and [x["qa_id"] for x in d["Data"][0]] == [0, 1, 2]
        assert [len(x["predicted_coi"]) for x in d["Data"][0]] == [0, 1, 2] and d["Data"][0][2]["coi"] == [0, 1]
        assert all(isinstance(x["prediction"], str) for x in d["Data"][0])
        ref[v] = d
    for name in outs:
        files = sorted(os.listdir(tmp_path / name))
        assert files == [f"clip_{v}.json" for v in range(4)], (name, files)
        for v in range(4):
            assert json.load(open(tmp_path / name / f"clip_{v}.json")) == ref[v], (name, v)


def test_bench_gpus_2_starts_its_own_ranks(dev):
    """`python bench.py --gpus 2` as a plain command (no torchrun around it): the parent starts two fresh rank
    processes before it touches the GPU, relays rank 0's JSON line and exits with their status. COGS_BENCH_REHEARSAL=1
    puts both ranks on this box's one GPU (gloo). N > 1 headline: strong scaling of the metric's 64-frame clip."""
    env = dict(ENV, COGS_BENCH_REHEARSAL="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=1100, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    assert d["config"]["frames"] == 64 and d["config"]["frames_per_gpu"] == 32
    assert d["cfg3"]["n_gpus"] == 2 and d["cfg3"]["frames_per_gpu"] == 128 and d["cfg3"]["scaling"] == "strong"
    assert d["weak"]["frames"] == 128 and d["weak"]["frames_per_gpu"] == 64


def test_bench_cfg5_replicas_mode_on_two_ranks(dev):
    """BASELINE configs[4] as a bench mode: `python bench.py --config cfg5 --gpus 2` -- every rank builds the whole model
    and answers its OWN 64-frame 480p clip through processor -> qa_selection -> generate (replicas only, no collective);
    value = frames of the answered clips per second of wall time. Rehearsal: both ranks on this box's one GPU."""
    env = dict(ENV, COGS_BENCH_REHEARSAL="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "cfg5", "--gpus", "2", "--steps", "1", "--warmup", "1",
                        "--decode-tokens", "8"], capture_output=True, text=True, timeout=1100, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["metric"] == base["metric"] and d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["scaling"] == "weak"
    assert d["config"]["frames"] == 128 and d["config"]["frames_per_gpu"] == 64 and "replicas only" in d["config"]["parallelism"]
    p = d["pipeline_rank0"]
    assert p["prompt_tokens"] == 15395 and p["new_tokens"] == 8 and p["kept_visual_tokens"] == 14784
    assert abs(d["value"] - 128 / (d["ms_per_step"] / 1e3)) < 0.01 * d["value"] and d["answers_per_s"] > 0
