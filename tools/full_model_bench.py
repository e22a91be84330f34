#!/usr/bin/env python3
"""Secondary end-to-end number: full MMBiDAF forward+backward at cfg2 sizes with a stub image embedder
and a fixed 10-step target, with a per-stage GPU time breakdown (embeddings+highway | hot path | decoder loop).
GPU box only."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn as nn

from mmbidaf_amd.model import MMBiDAF

dev = torch.device("cuda:0")
B, T, Ma, Mi, H = 32, 400, 256, 64, 100
Et, Ea, Ei = 300, 128, 1000
STEPS = 10
torch.manual_seed(224)


class StubBackbone(nn.Module):     # (N,3,h,w) -> (N,1000), stands in for the reference's ResNet (encoding.py:124)
    def __init__(self):
        super().__init__()
        self.fc = nn.Linear(3 * 8 * 8, 1000)

    def forward(self, x):
        return self.fc(nn.functional.adaptive_avg_pool2d(x, 8).flatten(1))


model = MMBiDAF(H, Et, Ea, Ei, dev, drop_prob=0.0, max_transcript_length=T + 5, image_backbone=StubBackbone()).to(dev)
model.train()
g = torch.Generator().manual_seed(1234)
text = torch.randn(B, T, Et, generator=g).to(dev)
audio = torch.randn(B, Ma, Ea, generator=g).to(dev)
images = torch.randn(B, Mi, 3, 32, 32, generator=g).to(dev)
tl, al, il = [T] * B, [Ma] * B, [Mi] * B
targets = torch.randint(0, T, (B, STEPS, 1), generator=g).float()
tlen = [STEPS] * B
params = [p for p in model.parameters() if p.requires_grad]


def step():
    for p in params:
        p.grad = None
    out, loss = model(text, tl, audio, al, images, il, targets, tlen, STEPS)
    loss.backward()
    return loss


for _ in range(3):
    step()
torch.cuda.synchronize()
N = 10
t0 = time.perf_counter()
for _ in range(N):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / N
print(f"full model fwd+bwd: {dt*1e3:.2f} ms/step  {B/dt:.1f} samples/s  (B={B}, T={T}, {STEPS} decode steps)")

# stage breakdown of the forward pass with events
ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
with torch.no_grad():
    model.eval()
    for rep in range(2):
        ev[0].record()
        te = model.emb(text); ae = model.a_emb(audio)
        ie = model.i_emb(model.image_keyframes_emb(images.view(-1, 3, 32, 32)).view(B, Mi, -1)) if hasattr(model, "image_keyframes_emb") else None
        ev[1].record()
        mod_a, hid_a, mod_i, hid_i, text_mask = model.hot_path(te, ae, ie, tl, al, il)
        ev[2].record()
        model.train()
        model.decode(text, T, mod_a, hid_a, mod_i, hid_i, text_mask, targets, STEPS)
        model.eval()
        ev[3].record()
        torch.cuda.synchronize()
    print(f"forward stages: embeddings+highway {ev[0].elapsed_time(ev[1]):.2f} ms | hot path {ev[1].elapsed_time(ev[2]):.2f} ms | "
          f"decoder ({STEPS} steps) {ev[2].elapsed_time(ev[3]):.2f} ms")
