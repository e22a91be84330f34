#!/usr/bin/env python3
"""Random-shape parity sweep of the fused attention against the CPU oracle (GPU box): N cases with random B, T, M, ragged prefix
lengths, with and without dropped copies; prints the worst relative error per output / gradient and fails on the first case
beyond tolerance.   python tools/att_fuzz.py [N] [seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import random

import torch

from mmbidaf_amd import functional as MF
from oracle import mmbidaf_oracle as O

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = random.Random(seed)
dev = torch.device("cuda:0")
worst = {}
for case in range(N):
    B = rng.choice([1, 2, 3, 8])
    T = rng.choice([1, 17, 32, 33, 63, 64, 65, 100, 129, 200, 257, 400])
    M = rng.choice([1, 9, 31, 32, 33, 64, 65, 97, 128, 160, 256, 300])
    D = rng.choice([200, 200, 200, 64, 208])
    drop = rng.random() < 0.35
    g = torch.Generator().manual_seed(seed * 100000 + case)
    text, mod = torch.randn(B, T, D, generator=g), torch.randn(B, M, D, generator=g)
    tl = [T] + [rng.randint(1, T) for _ in range(B - 1)]
    ml = [M] + [rng.randint(1, M) for _ in range(B - 1)]
    ps = [torch.randn(D, 1, generator=g) * 0.1, torch.randn(D, 1, generator=g) * 0.1, torch.randn(1, 1, D, generator=g) * 0.1, torch.randn(1, generator=g)]
    cot = torch.randn(B, T, 4 * D, generator=g)
    keep = ((torch.rand(B, T, D, generator=g) > 0.2).float() / 0.8, (torch.rand(B, M, D, generator=g) > 0.2).float() / 0.8) if drop else None

    def run(to):
        t_ = text.clone().to(to).requires_grad_(True)
        m_ = mod.clone().to(to).requires_grad_(True)
        p_ = [p.clone().to(to).requires_grad_(True) for p in ps]
        return t_, m_, p_
    # one-element softmaxes (M = 1 / T = 1): the gradients that are analytically zero come out as 3e-4 of ROUND-OFF in the fp32
    # reference (a sum over B*T terms of (1 - sum P2) * sum P2 g) where the kernels return 0 to the bit -- the reference runs in
    # float64 there, as tests/test_gpu_parity.py::test_attention_one_element_softmaxes_at_full_batch_vs_oracle does
    rdt = torch.float64 if (M == 1 or T == 1) else torch.float32
    t_, m_, p_ = run("cpu")
    t_, m_, p_ = t_.detach().to(rdt).requires_grad_(True), m_.detach().to(rdt).requires_grad_(True), [p.detach().to(rdt).requires_grad_(True) for p in p_]
    kw = dict(text_d=t_ * keep[0].to(rdt), mod_d=m_ * keep[1].to(rdt)) if drop else {}
    ref = O.bidaf_attention(t_, m_, O.get_mask(T, tl), O.get_mask(M, ml), *p_, **kw)
    (ref * cot.to(rdt)).sum().backward()
    tg, mg, pg = run(dev)
    tlt, mlt = torch.tensor(tl, dtype=torch.int32, device=dev), torch.tensor(ml, dtype=torch.int32, device=dev)
    drops = (tg * keep[0].to(dev), mg * keep[1].to(dev)) if drop else (None, None)
    out = MF.bidaf_attention_group([(tg, mg, MF.PrefixMask(tl, T, tlt), MF.PrefixMask(ml, M, mlt), *pg, *drops)])[0]
    (out * cot.to(dev)).sum().backward()
    torch.cuda.synchronize()
    pairs = [("out", out.detach().cpu(), ref.detach()), ("d_text", tg.grad.cpu(), t_.grad), ("d_mod", mg.grad.cpu(), m_.grad)] + \
            [(n, a.grad.cpu(), b.grad) for n, a, b in zip(("d_w_t", "d_w_m", "d_w_tm"), pg, p_)]
    for name, a, b in pairs:
        scale = max(1.0, b.abs().max().item())
        err = (a.double() - b.double()).abs().max().item() / scale
        lim = 1e-4
        if name.startswith("d_w"):                              # (parameter gradients: see the note at the end)
            lim = 3e-4
        if not torch.isfinite(a).all() or err > lim:
            print(f"FAIL case {case}: B={B} T={T} M={M} D={D} drop={drop} {name} err {err:.3e}")
            sys.exit(1)
        worst[name] = max(worst.get(name, 0.0), err)
print(f"{N} cases (seed {seed}) within 1e-4 of scale (parameter gradients 3e-4); worst: " + "  ".join(f"{k} {v:.2e}" for k, v in worst.items()))
# Parameter gradients get 3e-4 of scale (sums over B*T*M products).  One-element softmaxes (M = 1 / T = 1) no longer need an exception
# (round 5): the gradient sweeps take the gradient term of a softmax whose saved sum of exponentials is exactly 1 as exactly 0, as
# torch's softmax backward does, and leave the identically-zero halves out of the rank-1 sums (csrc/bidaf.hip, onehot_inv);
# tests/test_gpu_parity.py::test_attention_one_element_softmaxes_at_full_batch_vs_oracle holds those shapes to absolute 1e-4.
