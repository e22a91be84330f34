#!/usr/bin/env python3
"""cProfile of the BACKWARD half of the single-node region step (it runs on the autograd engine's thread, out of sight of a
profiler on the calling thread): the node's backward is wrapped before the first step.  GPU box only."""
import os, sys, cProfile, pstats, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mmbidaf_amd import synth, region_fn
from mmbidaf_amd.hot_region import HotRegion

pr = cProfile.Profile()
orig = region_fn._RegionFn.backward


def wrapped(ctx, *gs):
    pr.enable()
    try:
        return orig(ctx, *gs)
    finally:
        pr.disable()


region_fn._RegionFn.backward = staticmethod(wrapped)
dev = torch.device("cuda:0")
torch.manual_seed(224)
region = HotRegion(100).to(dev)
region.eval()
batch = synth.make_batch("cfg2", device=dev)
xs = [batch[k].requires_grad_(True) for k in ("x_text", "x_aud", "x_img")]
params = list(region.parameters())


def step():
    for p in params:
        p.grad = None
    for x in xs:
        x.grad = None
    outs = region(*xs, batch["text_len"], batch["aud_len"], batch["img_len"])
    synth.region_loss(outs, batch).backward()


for _ in range(10):
    step()
torch.cuda.synchronize()
pr.clear()
N = 100
for _ in range(N):
    step()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(30)
print(f"(totals over {N} steps; divide by {N})")
print(s.getvalue()[:7000])
