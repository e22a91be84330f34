#!/usr/bin/env python3
"""Development aid: run every HIP entry point against the oracle / goldens on the GPU box and
print the error of each tensor (no early exit), so one gpurun round trip tells everything.

    python tools/gpu_check.py [gemm] [att] [lstm] [model]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

from conftest import load_cases, load_flat
from oracle import mmbidaf_oracle as O
from mmbidaf_amd import functional as MF
from mmbidaf_amd.attention import BiDAFAttention
from mmbidaf_amd.encoding import RNNEncoder, encode_group

dev = torch.device("cuda:0")
FAILS = []


def report(name, got, ref, tol=1e-4):
    got = got.detach().float().cpu()
    ref = ref.detach().float().cpu()
    if got.shape != ref.shape:
        print(f"  {name:34s} SHAPE {tuple(got.shape)} vs {tuple(ref.shape)}")
        FAILS.append(name)
        return
    err = (got.double() - ref.double()).abs().max().item() if got.numel() else 0.0
    scale = max(1.0, ref.abs().max().item()) if ref.numel() else 1.0
    bad = not np.isfinite(got.numpy()).all() or err > tol * scale
    print(f"  {name:34s} maxerr {err:.3e}  scale {scale:.2e}  {'FAIL' if bad else 'ok'}")
    if bad:
        FAILS.append(name)


def check_gemm():
    print("== gemm")
    g = torch.Generator().manual_seed(0)
    for (M, N, K) in [(37, 29, 19), (128, 208, 64), (300, 400, 100), (1000, 100, 800), (800, 200, 1300), (64, 5, 7)]:
        for ta in (False, True):
            for tb in (False, True):
                a = torch.randn((K, M) if ta else (M, K), generator=g)
                b = torch.randn((N, K) if tb else (K, N), generator=g)
                bias = torch.randn(N, generator=g)
                ref = (a.t() if ta else a) @ (b.t() if tb else b) + bias
                got = MF.gemm(a.to(dev), b.to(dev), bias.to(dev), ta=ta, tb=tb)
                report(f"gemm {M}x{N}x{K} ta={int(ta)} tb={int(tb)}", got, ref, tol=2e-5)


def run_att(c, drop=None):
    text = c["text"].to(dev).requires_grad_(True)
    mod = c["mod"].to(dev).requires_grad_(True)
    ps = [c[k].to(dev).requires_grad_(True) for k in ("w_t", "w_m", "w_tm", "bias")]
    kw = {}
    if drop is not None:
        kw = dict(text_d=text * drop[0].to(dev), mod_d=mod * drop[1].to(dev))
    out = MF.bidaf_attention(text, mod, c["text_mask"].to(dev), c["mod_mask"].to(dev), *ps, **kw)
    (out * c["cot"].to(dev)).sum().backward()
    return out, text.grad, mod.grad, [p.grad for p in ps]


def check_att():
    print("== attention vs goldens")
    cases = load_cases("g3_bidaf_attention.npz")
    for name, c in cases.items():
        print(" case", name, tuple(c["text"].shape), tuple(c["mod"].shape))
        out, dt, dm, dps = run_att(c)
        report("out", out, c["out"])
        report("d_text", dt, c["d_text"])
        report("d_mod", dm, c["d_mod"])
        for k, gpar in zip(("d_w_t", "d_w_m", "d_w_tm"), dps):
            report(k, gpar, c[k])
        print(f"  d_bias got {dps[3].item():+.3e} ref {c['d_bias'].item():+.3e}")
    print("== attention vs oracle (dropout copies, ragged, cfg2-like)")
    g = torch.Generator().manual_seed(5)
    for (B, T, M, D, use_drop) in [(2, 50, 32, 200, True), (3, 70, 9, 200, False), (4, 400, 256, 200, False),
                                  (4, 400, 64, 200, True), (2, 33, 65, 64, False)]:
        text = torch.randn(B, T, D, generator=g)
        mod = torch.randn(B, M, D, generator=g)
        tl = torch.randint(1, T + 1, (B,), generator=g).tolist()
        ml = torch.randint(1, M + 1, (B,), generator=g).tolist()
        tl[0], ml[0] = T, M
        c = dict(text=text, mod=mod, text_mask=O.get_mask(T, tl), mod_mask=O.get_mask(M, ml),
                 cot=torch.randn(B, T, 4 * D, generator=g),
                 w_t=torch.randn(D, 1, generator=g) * 0.1, w_m=torch.randn(D, 1, generator=g) * 0.1,
                 w_tm=torch.randn(1, 1, D, generator=g) * 0.1, bias=torch.randn(1, generator=g))
        drop = None
        if use_drop:
            drop = ((torch.rand(B, T, D, generator=g) > 0.2).float() / 0.8, (torch.rand(B, M, D, generator=g) > 0.2).float() / 0.8)
        t_ = text.clone().requires_grad_(True)
        m_ = mod.clone().requires_grad_(True)
        ps = [c[k].clone().requires_grad_(True) for k in ("w_t", "w_m", "w_tm", "bias")]
        kw = dict(text_d=t_ * drop[0], mod_d=m_ * drop[1]) if use_drop else {}
        ref = O.bidaf_attention(t_, m_, c["text_mask"], c["mod_mask"], *ps, **kw)
        (ref * c["cot"]).sum().backward()
        print(f" case B{B} T{T} M{M} D{D} drop={use_drop}")
        out, dt, dm, dps = run_att(c, drop)
        report("out", out, ref)
        report("d_text", dt, t_.grad)
        report("d_mod", dm, m_.grad)
        for k, gpar, p in zip(("d_w_t", "d_w_m", "d_w_tm"), dps, ps):
            report(k, gpar, p.grad)
        print(f"  d_bias got {dps[3].item():+.3e} ref {ps[3].grad.item():+.3e}")


def _load_rnn(c, L, I, H):
    enc = RNNEncoder(I, H, L).to(dev)
    sd = {k[len("param__"):]: v for k, v in c.items() if k.startswith("param__")}
    enc.load_state_dict(sd)
    return enc


def check_lstm():
    print("== rnn encoder vs goldens")
    cases = load_cases("g4_rnn_encoder.npz")
    Ls = {"l1_ragged": 1, "l1_ties": 1, "l2_i8h": 2, "l1_full": 1, "l1_h100": 1, "l2_h25": 2}
    for name, c in cases.items():
        L = Ls[name]
        I = c["x"].shape[2]
        H = c["param__rnn.weight_hh_l0"].shape[1]
        print(" case", name, tuple(c["x"].shape), "H", H, "L", L, "len", c["lengths"].tolist())
        enc = _load_rnn(c, L, I, H)
        x = c["x"].to(dev).requires_grad_(True)
        y, hn = enc(x, c["lengths"].tolist())
        ((y * c["cot_y"].to(dev)).sum() + (hn * c["cot_h"].to(dev)).sum()).backward()
        report("y", y, c["y"])
        report("h_n", hn, c["h_n"])
        report("d_x", x.grad, c["d_x"])
        for n, p in enc.named_parameters():
            report("grad " + n, p.grad, c["grad__" + n])
    print("== rnn encoder vs oracle (H=100, grouped, cfg2-like lengths)")
    g = torch.Generator().manual_seed(9)
    encs, xs, lens, refs = [], [], [], []
    for (B, T, I, L) in [(4, 400, 100, 1), (4, 256, 100, 1), (4, 64, 100, 1)]:
        torch.manual_seed(100 + T)
        e = RNNEncoder(I, 100, L).to(dev)
        encs.append(e)
        xs.append(torch.randn(B, T, I, generator=g))
        l = torch.randint(T // 2, T + 1, (B,), generator=g).tolist()
        l[1] = T
        lens.append(l)
    xg = [x.to(dev).requires_grad_(True) for x in xs]
    outs = encode_group(encs, xg, lens)
    cots = [(torch.randn(*o[0].shape, generator=g), torch.randn(*o[1].shape, generator=g)) for o in outs]
    sum((y * cy.to(dev)).sum() + (h * ch.to(dev)).sum() for (y, h), (cy, ch) in zip(outs, cots)).backward()
    for e, x, l, (y, h), (cy, ch), xgi in zip(encs, xs, lens, outs, cots, xg):
        P = {k[4:]: v.detach().cpu().clone().requires_grad_(True) for k, v in e.state_dict().items()}
        xr = x.clone().requires_grad_(True)
        yr, hr = O.rnn_encoder(xr, l, P, e.rnn.num_layers)
        ((yr * cy).sum() + (hr * ch).sum()).backward()
        print(f" grouped T={x.shape[1]}")
        report("y", y, yr)
        report("h_n", h, hr)
        report("d_x", xgi.grad, xr.grad)
        for n, p in e.named_parameters():
            report("grad " + n, p.grad, P[n[4:]].grad)
    print("== 2-layer modelling encoder vs oracle (I=800,H=100)")
    torch.manual_seed(7)
    e = RNNEncoder(800, 100, 2).to(dev)
    x = torch.randn(3, 120, 800, generator=g) * 0.3
    l = [120, 77, 100]
    xd = x.to(dev).requires_grad_(True)
    y, h = e(xd, l)
    cy, ch = torch.randn(*y.shape, generator=g), torch.randn(*h.shape, generator=g)
    ((y * cy.to(dev)).sum() + (h * ch.to(dev)).sum()).backward()
    P = {k[4:]: v.detach().cpu().clone().requires_grad_(True) for k, v in e.state_dict().items()}
    xr = x.clone().requires_grad_(True)
    yr, hr = O.rnn_encoder(xr, l, P, 2)
    ((yr * cy).sum() + (hr * ch).sum()).backward()
    report("y", y, yr)
    report("h_n", h, hr)
    report("d_x", xd.grad, xr.grad)
    for n, p in e.named_parameters():
        report("grad " + n, p.grad, P[n[4:]].grad)


class _StubBackbone(torch.nn.Module):
    def __init__(self, w, b):
        super().__init__()
        self.fc = torch.nn.Linear(3, w.shape[0])
        with torch.no_grad():
            self.fc.weight.copy_(w)
            self.fc.bias.copy_(b)

    def forward(self, images):
        return self.fc(images.mean(dim=(2, 3)))


def check_model():
    print("== whole model vs G5 goldens")
    from mmbidaf_amd.model import MMBiDAF
    g = load_flat("g5_hot_region.npz")
    H, Et, Ea, Ei = 16, 24, 12, 20
    model = MMBiDAF(H, Et, Ea, Ei, dev, drop_prob=0.0, max_transcript_length=60,
                    image_backbone=_StubBackbone(g["resnet_w"], g["resnet_b"]))
    sd = {k[len("param__"):]: v for k, v in g.items() if k.startswith("param__")}
    missing = model.load_state_dict(sd, strict=False)
    print("  missing:", [k for k in missing.missing_keys if not k.startswith("image_keyframes_emb")],
          "unexpected:", missing.unexpected_keys)
    model.to(dev)
    tl, al, il = g["text_len"].tolist(), g["audio_len"].tolist(), g["image_len"].tolist()
    args = (g["text"].to(dev), tl, g["audio"].to(dev), al, g["images"].to(dev), il, g["targets"].to(dev), [4] * 3, 4)
    model.train()
    dist, loss = model(*args)
    report("train_dist", dist, g["train_dist"])
    report("train_loss", loss, g["train_loss"].reshape(()))
    model.zero_grad()
    loss.backward()
    for n, p in model.named_parameters():
        if ("grad__" + n) in g:
            report("grad " + n, p.grad if p.grad is not None else torch.zeros_like(p), g["grad__" + n])
    model.eval()
    with torch.no_grad():
        dist_e, loss_e = model(*args)
    report("eval_dist", dist_e, g["eval_dist"])
    report("eval_loss", loss_e, g["eval_loss"].reshape(()))


if __name__ == "__main__":
    which = sys.argv[1:] or ["gemm", "att", "lstm", "model"]
    t0 = time.time()
    for w in which:
        try:
            {"gemm": check_gemm, "att": check_att, "lstm": check_lstm, "model": check_model}[w]()
            torch.cuda.synchronize()
        except Exception as e:  # keep going: report everything in one trip
            import traceback
            traceback.print_exc()
            FAILS.append(w + ":exception")
    print(f"done in {time.time() - t0:.1f}s; FAILS = {len(FAILS)}")
    for f in FAILS[:60]:
        print("  FAIL", f)
    sys.exit(1 if FAILS else 0)
