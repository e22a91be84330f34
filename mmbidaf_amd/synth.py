"""Synthetic workload of the hot-path region: post-embedding features,
full or ragged lengths, fixed random cotangents.  No model keys: this is a workload, not a dataset."""
import torch

CONFIGS = {
    # name: (B, T_text, T_aud, T_img, H)
    "cfg1": (3, 50, 32, 8, 100),
    "cfg2": (32, 400, 256, 64, 100),
    "cfg4": (32, 1600, 1024, 256, 100),
    "cfg5": (64, 400, 256, 64, 512),     # BASELINE.json's H=512 case (lengths of cfg2): general-size kernels, fp32 arithmetic
}


def make_batch(cfg, rank=0, ragged=False, device="cpu", batch=None):
    """x ~ N(0,1) from manual_seed(1234 + rank); lengths full, or ~U{n/2..n} from seed 4321 + rank;
    cotangents R_a, R_i ~ N(0,1).  Returns dict of tensors / lists."""
    B, T, Ma, Mi, H = CONFIGS[cfg] if isinstance(cfg, str) else cfg
    if batch is not None:
        B = batch
    g = torch.Generator().manual_seed(1234 + rank)
    x_text = torch.randn(B, T, H, generator=g)
    x_aud = torch.randn(B, Ma, H, generator=g)
    x_img = torch.randn(B, Mi, H, generator=g)
    r_a = torch.randn(B, T, 2 * H, generator=g)
    r_i = torch.randn(B, T, 2 * H, generator=g)
    if ragged:
        gl = torch.Generator().manual_seed(4321 + rank)
        def lens(n):
            l = torch.randint(max(1, n // 2), n + 1, (B,), generator=gl).tolist()
            l[0] = n
            return l
        tl, al, il = lens(T), lens(Ma), lens(Mi)
    else:
        tl, al, il = [T] * B, [Ma] * B, [Mi] * B
    dev = torch.device(device)
    return dict(x_text=x_text.to(dev), x_aud=x_aud.to(dev), x_img=x_img.to(dev), r_a=r_a.to(dev), r_i=r_i.to(dev),
                text_len=tl, aud_len=al, img_len=il, B=B, T=T, Ma=Ma, Mi=Mi, H=H)


def region_loss(outs, batch):
    """<y_a,R_a> + <y_i,R_i> + sum(h_a) + sum(h_i)  (SURVEY 8(d))."""
    mod_a, hid_a, mod_i, hid_i = outs[:4]
    if mod_a.is_cuda:   # one launch forward, one backward (mmb_weighted_sums_*) instead of ~20 small torch kernels
        from . import functional as MF
        return MF.weighted_sums([mod_a, mod_i, hid_a, hid_i], [batch["r_a"], batch["r_i"], None, None])
    return (mod_a * batch["r_a"]).sum() + (mod_i * batch["r_i"]).sum() + hid_a.sum() + hid_i.sum()


def attention_algorithmic_bytes(B, T, M, D, backward=False):
    """Algorithmic HBM bytes of one fused attention: fwd 4B(5TD+MD), bwd 4B(6TD+2MD)."""
    return 4 * B * ((6 * T * D + 2 * M * D) if backward else (5 * T * D + M * D))
