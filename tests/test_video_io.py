"""Video front-end (cogstream_amd/video_io.py) against tests/golden/video_io.json: the reference's load_video /
_load_multimodal_data arithmetic (model/processing_cogreasoner.py:326-509) run with the decoder replaced by
video_io.select_frames (tests/golden/make_golden.py::golden_video_io) -- timestamps, durations, subsampling, padding,
per-content windows and the running offset between segments are compared exactly."""
import copy
import json
import os

import numpy as np

from cogstream_amd import video_io as vio

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "video_io.json")))


def _videos():
    vids = {}
    for k, (n, f, st, du) in G["specs"].items():
        fr = np.zeros((int(n), 4, 6, 3), np.uint8)
        fr[:, 0, 0, 0] = np.arange(int(n)) % 256
        vids[k] = vio.DecodedVideo(fr, f, st, du)
    return vids


def test_load_video_matches_reference_arithmetic():
    vids = _videos()
    for case in G["load_video"]:
        kw = dict(case["args"])
        frames, ts, dur = vio.load_video(vids[kw.pop("video_path")], **kw)
        assert [int(f[0, 0, 0]) for f in frames] == case["frame_ids"], case["args"]
        assert ts == case["timestamps"], case["args"]
        assert dur == case["duration"]


def test_segments_are_stitched_like_the_reference():
    new_conv, all_ts = vio.load_multimodal_data(copy.deepcopy(G["conversation"]), _videos())
    segs = [c for m in new_conv if isinstance(m["content"], list) for c in m["content"]
            if isinstance(c, dict) and c.get("type") == "video"]
    assert len(segs) == len(G["segments"])
    for got, want in zip(segs, G["segments"]):
        assert got["num_frames"] == want["num_frames"] and got["timestamps"] == want["timestamps"]
        assert [int(f[0, 0, 0]) for f in got["video"]] == want["frame_ids"]
    assert all_ts == G["all_timestamps"]
    # the stitched conversation is what CogStreamProcessor takes: uint8 frames + timestamps per video content
    assert segs[0]["video"].dtype == np.uint8 and segs[0]["video"].shape[1:] == (4, 6, 3)
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from toy_tokenizer import ToyTokenizer
    from cogstream_amd.processing import CogStreamProcessor
    out = CogStreamProcessor(ToyTokenizer())(conversation=new_conv, add_system_prompt=True, add_generation_prompt=True,
                                              return_tensors="pt")
    assert out["all_timestamps"] == all_ts and out["total_image_num"] == len(all_ts)
    assert out["grid_sizes"][:, 0].tolist() == [s["num_frames"] for s in G["segments"]]


def test_fps_filter_round_up_slots():
    v = vio.DecodedVideo(np.zeros((10, 2, 2, 3), np.uint8), native_fps=4.0)      # pts 0, .25, ... 2.25
    # slot = ceil(t * fps): frame 0 -> 0, frames 1-4 -> 1, frames 5-8 -> 2, frame 9 -> 3; a slot shows its latest frame
    assert vio.select_frames(v, 2.5, False, 0.0, 1.0).tolist() == [0, 4, 8, 9]
    assert vio.select_frames(v, 1.0, True, 0.0, 1.0).tolist() == [0, 3]          # -t 1.0 keeps pts < 1.0
    assert vio.select_frames(v, 2.5, False, 0.0, None).tolist() == list(range(10))
