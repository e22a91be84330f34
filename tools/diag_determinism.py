#!/usr/bin/env python3
"""Diagnostic (GPU box): run the same cfg4 (or DIAG_CFG) region step N times and compare every output / gradient with the first
run.  Everything except sums of atomics (attention parameter gradients, K-split GEMM outputs) must repeat bit for bit."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mmbidaf_amd import synth, region_fn
from mmbidaf_amd.hot_region import HotRegion

cfg = os.environ.get("DIAG_CFG", "cfg4")
N = int(os.environ.get("DIAG_N", "20"))
d = torch.device("cuda:0")
torch.manual_seed(224)
B, T, Ma, Mi, H = synth.CONFIGS[cfg]
region = HotRegion(H).to(d)
batch = synth.make_batch(cfg, ragged=True, device=d)
for fn in (True, False):
    region_fn._ENABLED = fn
    first = None
    worst = {}
    for it in range(N):
        for p in region.parameters():
            p.grad = None
        xs = [batch[k].detach().clone().requires_grad_(True) for k in ("x_text", "x_aud", "x_img")]
        outs = region(*xs, batch["text_len"], batch["aud_len"], batch["img_len"])
        synth.region_loss(outs, batch).backward()
        torch.cuda.synchronize()
        cur = {"out%d" % i: o.detach().clone() for i, o in enumerate(outs)}
        cur.update({"d_x_text": xs[0].grad, "d_x_aud": xs[1].grad, "d_x_img": xs[2].grad})
        cur.update({"g:" + n: p.grad.clone() for n, p in region.named_parameters()})
        if first is None:
            first = cur
            continue
        for k in cur:
            e = (cur[k] - first[k]).abs().max().item()
            if e > worst.get(k, (0.0, -1))[0]:
                worst[k] = (e, it)
                if k.startswith("d_x") and e > 1e-6:
                    diff = (cur[k] - first[k]).abs()
                    per = diff.flatten(1).amax(1)
                    s = int(per.argmax())
                    tt = diff[s].amax(-1)
                    nz = (tt > 1e-7).nonzero().flatten()
                    print(f"  [{'node' if fn else 'modular'}] iteration {it}: {k} differs by {e:.3e}; worst sample {s}, rows with diff: {nz.numel()} "
                          f"(first {nz[:8].tolist()} last {nz[-3:].tolist()}), samples affected {(per > 1e-7).nonzero().flatten().tolist()}")
    print(f"{'single node' if fn else 'modular'} path, {N} runs: tensors that did not repeat exactly:")
    for k, (e, it) in sorted(worst.items(), key=lambda kv: -kv[1][0]):
        if e > 0:
            print(f"    {k:45s} max |run - first| = {e:.3e} (iteration {it})  max|first| = {first[k].abs().max().item():.3e}")
