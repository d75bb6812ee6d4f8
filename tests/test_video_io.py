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


def test_driver_loop_over_decoded_segment_files_on_a_stub_model(tmp_path):
    """answer_generate.answer_video = the per-video body of the reference's inference() (evaluate/answer_generate.py:
    106-148): segments grouped by Event_Time and zipped with the naturally sorted segment files, every question
    answered with the whole conversation so far, records in the reference's result format. The clips are decoded-frame
    files (video_io.write_decoded_video) that the processor samples at 1 fps / max_frames 180 with timestamps stitched
    across segments. The model is a stub (CPU): what is checked is the driver and the processor, not the kernels."""
    import json
    import os
    import sys
    import numpy as np
    import torch
    sys.path.insert(0, os.path.dirname(__file__))
    from toy_tokenizer import ToyTokenizer
    from cogstream_amd import processing as pr
    from cogstream_amd.answer_generate import VideoDataset, answer_video, natural_sort_segments, save_to_json
    from cogstream_amd.video_io import read_decoded_video, write_decoded_video

    vdir, qdir = tmp_path / "videos", tmp_path / "queries"
    (vdir / "clipA").mkdir(parents=True)
    qdir.mkdir()
    for s, n in ((0, 6), (1, 8), (10, 4)):        # segment_10 sorts after segment_1 (natural order), 2 fps native
        fr, _ = pr.synthetic_clip(n, 56, 84, kind="drift", clip_idx=s)
        write_decoded_video(str(vdir / "clipA" / f"segment_{s}.npz"), fr, native_fps=2.0)
    assert natural_sort_segments(str(vdir / "clipA")) == ["segment_0.npz", "segment_1.npz", "segment_10.npz"]
    dv = read_decoded_video(str(vdir / "clipA" / "segment_1.npz"))
    assert dv.frames.shape == (8, 56, 84, 3) and dv.native_fps == 2.0
    chain = [{"Q": "q0?", "A": "a0", "info": {"Event_Time": 3, "relevance": []}},
             {"Q": "q1?", "A": "a1", "info": {"Event_Time": 3, "relevance": [1]}},
             {"Q": "q2?", "A": "a2", "info": {"Event_Time": 7, "relevance": [0, 1]}},
             {"Q": "q3?", "A": "a3", "info": {"Event_Time": 9, "relevance": [0, 0, 1]}}]
    json.dump([chain], open(qdir / "clipA.json", "w"))
    json.dump([chain], open(qdir / "missing_video.json", "w"))
    ds = VideoDataset(str(vdir), str(qdir))
    assert len(ds) == 1 and ds[0]["video_path"].endswith("clipA") and ds[0]["query_chain"] == chain

    class Stub:
        device, dtype, _adapters = torch.device("cpu"), torch.float32, {}
        calls = []

        def qa_selection(self, **kw):
            n = len(kw["hist_qs"])
            self.calls.append((n, kw["pixel_values"].shape[0], list(kw["all_timestamps"]), kw["current_question"]))
            sel = "" if n == 0 else "[" + ",".join(["yes"] + [str(i) for i in range(0, n, 2)]) + "]"
            return {**kw, "new_input_ids": kw["input_ids"], "new_attention_mask": kw["attention_mask"],
                    "selection_module_output": sel, "if_visual": True}

        def generate(self, **kw):
            return torch.tensor([[ord("o"), ord("k"), 48 + len(kw["hist_qs"])]]), kw["selection_module_output"]

    model = Stub()
    data = answer_video(model, pr.CogStreamProcessor(ToyTokenizer()), ds[0]["video_path"], ds[0]["query_chain"], max_new_tokens=3)
    recs = data[0]
    assert [r["qa_id"] for r in recs] == [0, 1, 2, 3] and [r["prediction"] for r in recs] == ["ok0", "ok1", "ok2", "ok3"]
    assert [r["predicted_coi"] for r in recs] == [[], [1], [1, 0], [1, 0, 1]] and [r["coi"] for r in recs] == [[], [1], [0, 1], [0, 0, 1]]
    assert [r["answer"] for r in recs] == ["a0", "a1", "a2", "a3"] and all(r["predicted_visual"] for r in recs)
    # the conversation grows by one clip per segment: 4 frames (6 at 2 fps -> 1 fps, round up), then + 4, then + 2; the 1 fps
    # timestamp grid continues across the segments
    n_hist, n_patch, ts, q = zip(*model.calls)
    assert n_hist == (0, 1, 2, 3) and q == ("q0?", "q1?", "q2?", "q3?")
    assert [len(t) for t in ts] == [4, 4, 8, 10] and list(ts[3]) == sorted(ts[3]) and ts[3][:4] == ts[0]
    assert len(set(ts[3])) == 10
    path = save_to_json("clipA", data, str(tmp_path / "out"))
    assert json.load(open(path)) == {"video_name": "clipA", "Data": data}
