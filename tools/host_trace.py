#!/usr/bin/env python3
"""Where the HOST spends a region step: wall time inside every autograd Function forward / backward of the package and inside
the ctypes library calls, per step (GPU box).  Compare with the GPU time of the step (bench.py)."""
import os, sys, time, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mmbidaf_amd import synth, functional as MF, _lib
from mmbidaf_amd.hot_region import HotRegion

acc = collections.defaultdict(lambda: [0.0, 0])
def wrap(obj, name, label):
    fn = getattr(obj, name)
    def w(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            acc[label][0] += time.perf_counter() - t0
            acc[label][1] += 1
    setattr(obj, name, staticmethod(w) if isinstance(obj, type) else w)

lib = _lib.load()
for n in ("mmb_bilstm_layer_fwd", "mmb_bilstm_layer_bwd_phase", "mmb_bilstm_layer_bwd", "mmb_bidaf_group_fwd", "mmb_bidaf_group_bwd",
          "mmb_weighted_sums_fwd", "mmb_weighted_sums_bwd"):
    wrap(lib, n, "C " + n)
for cls in (MF._BiLSTMLayerFn, MF._BiDAFAttentionGroupFn, MF._WeightedSumsFn):
    wrap(cls, "forward", cls.__name__ + ".forward")
    wrap(cls, "backward", cls.__name__ + ".backward")

dev = torch.device("cuda:0")
torch.manual_seed(224)
region = HotRegion(100).to(dev)
batch = synth.make_batch("cfg2", device=dev)
xs = [batch[k].requires_grad_(True) for k in ("x_text", "x_aud", "x_img")]
params = list(region.parameters())
def step():
    for p in params: p.grad = None
    for x in xs: x.grad = None
    t0 = time.perf_counter()
    outs = region(*xs, batch["text_len"], batch["aud_len"], batch["img_len"])
    loss = synth.region_loss(outs, batch)
    t1 = time.perf_counter()
    loss.backward()
    t2 = time.perf_counter()
    acc["step forward (host)"][0] += t1 - t0; acc["step forward (host)"][1] += 1
    acc["step backward (host)"][0] += t2 - t1; acc["step backward (host)"][1] += 1
for _ in range(5): step()
torch.cuda.synchronize()
acc.clear()
N = 20
t0 = time.perf_counter()
for _ in range(N): step()
torch.cuda.synchronize()
wall = time.perf_counter() - t0
print(f"wall {wall / N * 1e3:.3f} ms/step")
for k, (t, n) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    print(f"{k:45s} {t / N * 1e3:8.3f} ms/step  {n / N:5.1f} calls/step  {t / max(n, 1) * 1e6:8.1f} us/call")
