#!/usr/bin/env python3
"""Diagnostic (GPU box): cfg4 at full size, GPU vs the CPU oracle, per sample and per stage of the audio branch of the backward
pass: the cotangent of the audio encoder's output (= d_mod of the text<->audio attention) and d_x_aud."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mmbidaf_amd import synth, region_fn
from mmbidaf_amd.hot_region import HotRegion
from mmbidaf_amd.attention import BiDAFAttention
from oracle import mmbidaf_oracle as O

B = int(os.environ.get("DIAG_B", "32"))
d = torch.device("cuda:0")
shape = (B, 1600, 1024, 256, 100)
torch.manual_seed(224)
region = HotRegion(100).to(d)
batch = synth.make_batch(shape, ragged=True)
gpu = {k: (v.to(d) if torch.is_tensor(v) else v) for k, v in batch.items()}

FN = os.environ.get("DIAG_FN") == "1"   # 1: the single-node path (no hooks on the encoder outputs)
region_fn._ENABLED = FN
cap = []
orig = BiDAFAttention.forward_group
def fg(mods, texts, modalities, tms, mms):
    for m in modalities:
        m.retain_grad()
    texts[0].retain_grad()
    cap.append((texts[0], modalities[0], modalities[1]))
    return orig(mods, texts, modalities, tms, mms)
BiDAFAttention.forward_group = staticmethod(fg)
xs = [gpu[k].requires_grad_(True) for k in ("x_text", "x_aud", "x_img")]
outs = region(*xs, batch["text_len"], batch["aud_len"], batch["img_len"])
synth.region_loss(outs, gpu).backward()
torch.cuda.synchronize()
g_te, g_ae, g_ie = [t.grad.cpu() for t in cap[0]] if cap else (None, None, None)

torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
ref = O.HotRegionCPU(region.state_dict(), 100)
keep = {}
orig_enc = O.rnn_encoder_aten
def enc(x, lengths, rnn, *a, **k):
    y, h = orig_enc(x, lengths, rnn, *a, **k)
    if y.shape[2] == 200 and x.shape[2] == 100:
        y.retain_grad()
        keep[x.shape[1]] = y
    return y, h
O.rnn_encoder_aten = enc
xr = [batch[k].clone().requires_grad_(True) for k in ("x_text", "x_aud", "x_img")]
routs = ref(*xr, batch["text_len"], batch["aud_len"], batch["img_len"])
synth.region_loss(routs, batch).backward()
stages = [("d text_enc out", g_te, keep[1600].grad), ("d audio_enc out", g_ae, keep[1024].grad), ("d image_enc out", g_ie, keep[256].grad)] if cap else []
for n_, a_, b_ in zip(("mod_a", "hid_a", "mod_i", "hid_i", "dec"), outs, routs):
    print(f"{n_:16s} max err {(a_.detach().cpu() - b_.detach()).abs().max().item():.3e}")
rg = ref.named_grads()
for n_, p_ in region.named_parameters():
    if "audio_enc" in n_ or "bidaf_att_audio" in n_:
        print(f"grad {n_:40s} err {(p_.grad.cpu() - rg[n_]).abs().max().item():.3e} max|ref| {rg[n_].abs().max().item():.3e}")
for name, got, want in stages + [("d_x_text", xs[0].grad.cpu(), xr[0].grad), ("d_x_aud", xs[1].grad.cpu(), xr[1].grad), ("d_x_img", xs[2].grad.cpu(), xr[2].grad)]:
    err = (got - want).abs()
    per = err.flatten(1).amax(1)
    worst = int(per.argmax())
    t_w = int(err[worst].amax(-1).argmax())
    print(f"{name:16s} max err {err.max().item():.3e}  max|ref| {want.abs().max().item():.3e}  worst sample {worst} (err {per[worst].item():.3e}) at t = {t_w};  samples with err > 1e-4: {(per > 1e-4).nonzero().flatten().tolist()}")
print("aud_len", batch["aud_len"])
print("text_len", batch["text_len"])
w = int((xs[1].grad.cpu() - xr[1].grad).abs().flatten(1).amax(1).argmax())
e = (xs[1].grad.cpu() - xr[1].grad)[w].abs().amax(-1)
print("worst sample", w, "len", batch["aud_len"][w], "error by t (every 64):", [f"{v:.1e}" for v in e[::64].tolist()])
print("ref |d_x_aud| by t (every 64):", [f"{v:.1e}" for v in xr[1].grad[w].abs().amax(-1)[::64].tolist()])
print("ref |d audio_enc out| by t (every 64):", [f"{v:.1e}" for v in keep[1024].grad[w].abs().amax(-1)[::64].tolist()])
