"""bench.py's contract: the FLOP model behind `roofline.achieved` reproduces SURVEY.md section 8(d)'s per-frame figures
(CPU), and one tiny run prints ONE JSON line with the keys the driver reads, BASELINE.json's metric, the roofline
object and (when asked for) the CPU baseline (GPU)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_flop_model_matches_survey_figures():
    import bench
    from cogstream_amd.weights import VisionConfig
    cfg = VisionConfig()
    # cfg2: 64 frames of 22 x 42 patches (924 per frame), 231 merged tokens per frame, projector to 3584
    per_frame = (bench.vit_gemm_flops(59136, 14784, cfg, 3584) + bench.vit_attn_flops(64, 924, cfg)) / 64
    assert abs(per_frame / 1e9 - (867 + 0.50e3 / 64)) < 0.01 * 875       # 867 GFLOP/frame encoder + projector share
    # cfg1: 8 frames of 16 x 16 patches -> 219 GFLOP/frame; cfg3: 256 frames of 10 x 20 -> 170 GFLOP/frame
    enc1 = (bench.vit_gemm_flops(2048, 512, cfg, 3584) + bench.vit_attn_flops(8, 256, cfg)) / 8
    enc3 = (bench.vit_gemm_flops(51200, 12800, cfg, 3584) + bench.vit_attn_flops(256, 200, cfg)) / 256
    proj = 2.0 * (1152 * 3584 + 3584 * 3584)
    assert abs((enc1 - 64 * proj) / 1e9 - 219) < 3 and abs((enc3 - 50 * proj) / 1e9 - 170) < 3


def test_bench_tokenizer_reproduces_the_real_prompt_lengths():
    """bench.PromptLengthTokenizer stands in for the checkpoint's Qwen2 tokenizer (whose vocabulary does not ship) in the
    `pipeline` / cfg5 timings: its token COUNTS on the bench's prompts equal the real tokenizer's, as recorded in
    tests/golden/tokenizer.json (cfg2 15 395, cfg1 621, cfg3 15 295 prompt tokens; event-summary prompt 974)"""
    import torch
    import bench
    from cogstream_amd import processing as pr
    from cogstream_amd.chat import create_visual_summary_prompt
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "tokenizer.json")))
    tok = bench.PromptLengthTokenizer()
    assert [p["len"] for p in fx["prompts"]] == [15395, 621, 15295]
    for p in fx["prompts"]:
        T, P = p["T"], p["P"]
        conv = [{"role": "user", "content": [{"type": "video", "num_frames": T, "timestamps": [float(i) for i in range(T)]},
                                             {"type": "text", "text": p["question"]}]}]
        text = pr.expand_image_tokens(pr.render_conversation(conv, True, True), [P] * T)
        ids = tok(text)["input_ids"]
        assert ids.shape == (1, p["len"]) and bench.prompt_tokens(T, P) == p["len"]
        assert int((ids == fx["image_token_id"]).sum()) == T * P and int(ids[0, 0]) == fx["im_start"]
    sp = create_visual_summary_prompt(15 * 50, torch.arange(15, dtype=torch.float32) + 30)
    assert tok(sp)["input_ids"].shape[1] == fx["summary_prompt_len"] == 974


def test_bench_gpus_n_launches_its_own_ranks_without_touching_the_gpu(monkeypatch, capsys):
    """`python bench.py --gpus 4` outside torch.distributed.run: the parent builds the launcher command for 4 fresh rank
    processes with the same flags on 127.0.0.1, relays rank 0's JSON line and returns their status"""
    import io
    import subprocess
    import bench
    seen = {}

    class FakeProc:
        def __init__(self, cmd, **kw):
            seen["cmd"], seen["env"] = cmd, kw["env"]
            self.stdout = io.StringIO("noise from a rank\n{\"metric\": \"m\", \"n_gpus\": 4}\n")

        def wait(self):
            return 3

    monkeypatch.setattr(subprocess, "Popen", FakeProc)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2"])
    assert bench.launch_ranks(4) == 3
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "2"]
    assert cmd[-5].endswith("bench.py") and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    out = capsys.readouterr()
    assert out.out.strip() == '{"metric": "m", "n_gpus": 4}' and "noise from a rank" in out.err


@pytest.mark.gpu
def test_bench_prints_one_json_line_with_the_contract_keys():
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--frames", "8", "--steps", "1", "--warmup", "1",
                        "--no-llm", "--no-cpu"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["metric"] == base["metric"] and d["unit"] == "frames/s"
    for k in ("value", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["dtype"] == "bf16" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and rf["peak"] == 2500.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and rf["achieved"] > 0
    assert abs(d["value"] - 8 / (d["ms_per_step"] / 1e3)) < 0.01 * d["value"]
