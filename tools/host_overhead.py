#!/usr/bin/env python3
"""How long does the host need to ISSUE one fwd+bwd step of the region vs. how long the GPU needs to run it?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mmbidaf_amd import synth
from mmbidaf_amd.hot_region import HotRegion
dev = torch.device("cuda:0")
torch.manual_seed(224)
region = HotRegion(100).to(dev)
batch = synth.make_batch("cfg2", device=dev)
xs = [batch[k].requires_grad_(True) for k in ("x_text", "x_aud", "x_img")]
params = list(region.parameters())
def step():
    for p in params: p.grad = None
    for x in xs: x.grad = None
    outs = region(*xs, batch["text_len"], batch["aud_len"], batch["img_len"])
    synth.region_loss(outs, batch).backward()
for _ in range(5): step()
torch.cuda.synchronize()
N = 20
t0 = time.perf_counter()
for _ in range(N): step()
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"host issue {t_issue/N*1e3:.2f} ms/step, wall {t_all/N*1e3:.2f} ms/step")
# forward only / backward only split of host time
t0 = time.perf_counter()
for _ in range(N):
    outs = region(*xs, batch["text_len"], batch["aud_len"], batch["img_len"])
    loss = synth.region_loss(outs, batch)
t_f = time.perf_counter() - t0
torch.cuda.synchronize()
print(f"host issue forward only {t_f/N*1e3:.2f} ms")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(5): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
