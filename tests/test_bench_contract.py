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
