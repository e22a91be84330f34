#!/usr/bin/env python3
"""cProfile of the host side of eager region steps (GPU box): where the Python time of the single-node path goes."""
import os, sys, cProfile, pstats, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mmbidaf_amd import synth
from mmbidaf_amd.hot_region import HotRegion
dev = torch.device("cuda:0")
torch.manual_seed(224)
drop = float(os.environ.get("DROP", "0"))
region = HotRegion(100, drop_prob=drop).to(dev)
region.train(drop > 0)
batch = synth.make_batch("cfg2", device=dev)
xs = [batch[k].requires_grad_(True) for k in ("x_text", "x_aud", "x_img")]
params = list(region.parameters())
def step():
    for p in params: p.grad = None
    for x in xs: x.grad = None
    outs = region(*xs, batch["text_len"], batch["aud_len"], batch["img_len"])
    synth.region_loss(outs, batch).backward()
for _ in range(10): step()
torch.cuda.synchronize()
N = 100
pr = cProfile.Profile()
pr.enable()
for _ in range(N): step()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
st = pstats.Stats(pr, stream=s).sort_stats("tottime")
st.print_stats(28)
txt = s.getvalue()
# per-step microseconds
print(f"(times below are totals over {N} steps; divide by {N})")
print(txt[:6000])
