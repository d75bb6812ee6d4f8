"""Seeded input construction shared by make_golden.py (reference side, build container) and the parity
tests (both boxes), so the e2e fixtures only need to store outputs + an input checksum."""
import torch

GH = GW = 4
P = 4


def e2e_inputs(tag: str):
    T, hist = {"a": (8, True), "b": (150, False), "c": (150, False)}[tag]
    g = torch.Generator().manual_seed(41 + T)
    pf = GH * GW
    base = torch.rand(1, pf, 588, generator=g) * 2 - 1
    pix = base.repeat(T, 1, 1)
    # piecewise-constant "scenes" + small noise: events are well separated, pixel-diff prunes some tokens
    scene = (torch.arange(T) // 15).float()
    pix = pix + 0.6 * torch.sin(scene[:, None, None] * torch.arange(1, 589)[None, None, :] * 0.37)
    # per-frame noise scale 1x..15x inside each scene: distances to the event centroid are well separated, so
    # the "2 frames nearest the centroid" pick (cogreasoner_chat.py:50-64) does not hinge on rounding
    noise = 0.004 * torch.randn(T, pf, 588, generator=g) * (1 + (torch.arange(T) % 15))[:, None, None]
    pix = (pix + noise).reshape(-1, 588)
    pix[pf * 5:pf * 6] = pix[pf * 4:pf * 5]                    # an exactly repeated frame
    pix[pf * 7:pf * 7 + 8] = pix[pf * 6:pf * 6 + 8]            # half of frame 7 unchanged (2 of 4 merged tokens)
    grid, merge = torch.tensor([[T, GH, GW]]), torch.tensor([2])
    ts = [float(i) for i in range(T)]
    img = "<image>" * P
    sys_ = "<|im_start|>system\nYou are a helpful assistant.<|im_end|>\n"
    if hist:
        hq, ha, cur = ["What is on the table?", "Who enters?"], ["A red cup.", "A man."], "What does he pick up?"
        half = ",".join(f"Time {t:.1f}s:{img}" for t in ts[:4])
        half2 = ",".join(f"Time {t:.1f}s:{img}" for t in ts[4:])
        text = (sys_ + f"<|im_start|>user\n{half}\n{hq[0]}<|im_end|>\n<|im_start|>assistant\n{ha[0]}<|im_end|>\n"
                f"<|im_start|>user\n{half2}\n{hq[1]}<|im_end|>\n<|im_start|>assistant\n{ha[1]}<|im_end|>\n"
                f"<|im_start|>user\n{cur}<|im_end|>\n<|im_start|>assistant\n")
    else:
        hq, ha, cur = [], [], "What is happening in the video?"
        frames = ",".join(f"Time {t:.1f}s:{img}" for t in ts)
        text = sys_ + f"<|im_start|>user\n{frames}\n{cur}<|im_end|>\n<|im_start|>assistant\n"
    return dict(T=T, pixel_values=pix, grid_sizes=grid, merge_sizes=merge, timestamps=ts, text=text, hist_qs=hq,
                hist_as=ha, current_question=cur)


FORCED_COSINE = [0.9, 0.2, 0.8, 0.1, 0.3, 0.95, 0.44, 0.46, 0.7, 0.05]
