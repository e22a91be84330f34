import sys, torch
sys.path.insert(0, "/root/repo")
from mmbidaf_amd import synth, region_fn, _lib
from mmbidaf_amd.hot_region import HotRegion
d = torch.device("cuda:0")
shape = (16, 300, 190, 40, 100)
torch.manual_seed(224)
region = HotRegion(100).to(d).eval()
batch = synth.make_batch(shape, ragged=True)
gpu = {k: (v.to(d) if torch.is_tensor(v) else v) for k, v in batch.items()}
region_fn._FWD_STREAM_MIN_ROWS = 0
def run(cfg):
    region_fn._FWD_STREAM = cfg
    for p in region.parameters(): p.grad = None
    xs = [gpu[k].detach().clone().requires_grad_(True) for k in ("x_text", "x_aud", "x_img")]
    outs = region(*xs, batch["text_len"], batch["aud_len"], batch["img_len"])
    synth.region_loss(outs, gpu).backward()
    torch.cuda.synchronize()
    return {n: p.grad.clone() for n, p in region.named_parameters()}
a = run([(8,1),(8,3),(8,1)]); b = run(None); b2 = run(None)
for n in a:
    e = (a[n]-b[n]).abs().max().item(); e2 = (b2[n]-b[n]).abs().max().item()
    if e or e2: print(n, "streamed-vs-classic", e, "classic-vs-classic", e2, "max", b[n].abs().max().item())
