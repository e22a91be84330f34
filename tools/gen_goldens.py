#!/usr/bin/env python3
"""Generate the golden parity fixtures under tests/golden/ by RUNNING the reference.

Runs only in the build container (where /root/reference exists). The reference is imported
from where it lies (sys.path), with PYTHONDONTWRITEBYTECODE semantics, and nothing of it is
copied: only numeric inputs/outputs are stored (fixtures G1-G6).

The reference imports `torchvision` at module top (layers/attention.py:3, layers/encoding.py:3)
but uses it only in ImageEmbedding.__init__ (layers/encoding.py:124); torchvision is absent
from this image, so an in-memory stand-in module is registered before the import.  For the
whole-model fixture G5 the stand-in's `models.resnet101` returns a tiny frozen linear
"image embedder" (N,3,h,w)->(N,20) whose weight is stored in the fixture.

    python tools/gen_goldens.py            # writes tests/golden/*.npz + state_dict_keys.json
"""
import json
import os
import sys
import types

sys.dont_write_bytecode = True
import numpy as np
import torch
import torch.nn as nn

REF = os.environ.get("MMB_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
TESTS_DIR = os.path.dirname(OUT)                      # tests/: golden_recipe.py
OUT = os.environ.get("MMB_GOLDEN_OUT", OUT)           # (write somewhere else, e.g. to check that the committed fixtures reproduce)


class _StubResNet(nn.Module):
    """Stand-in for torchvision.models.resnet101: mean-pool + linear to 20 features."""

    def __init__(self):
        super().__init__()
        g = torch.Generator().manual_seed(77)
        self.fc = nn.Linear(3, 20)
        with torch.no_grad():
            self.fc.weight.copy_(torch.randn(20, 3, generator=g) * 0.5)
            self.fc.bias.copy_(torch.randn(20, generator=g) * 0.1)

    def forward(self, images):
        return self.fc(images.mean(dim=(2, 3)))


def _install_torchvision_stub():
    tv = types.ModuleType("torchvision")
    tvm = types.ModuleType("torchvision.models")
    tvm.resnet101 = lambda pretrained=True: _StubResNet()
    tv.models = tvm
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.models"] = tvm


def _np(t):
    return t.detach().cpu().numpy().copy()


def _save(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("wrote", path, os.path.getsize(path), "bytes")


def gen_masked_softmax(ref_att):
    """G1: masked_softmax, reference layers/attention.py:78-98."""
    g = torch.Generator().manual_seed(1)
    out = {}
    x = torch.randn(2, 6, 5, generator=g)
    # prefix mask on dim 2
    m2 = (torch.arange(5)[None, None, :] < torch.tensor([3, 5])[:, None, None])
    # non-prefix mask on dim 1, with one fully masked batch entry
    m1 = torch.tensor([[1, 0, 1, 1, 0, 1], [0, 0, 0, 0, 0, 0]], dtype=torch.bool)[:, :, None]
    out["x"] = _np(x)
    out["mask_dim2"] = _np(m2)
    out["mask_dim1"] = _np(m1)
    out["y_dim2"] = _np(ref_att.masked_softmax(x, m2, dim=2))
    out["y_dim1"] = _np(ref_att.masked_softmax(x, m1, dim=1))
    out["y_dim2_log"] = _np(ref_att.masked_softmax(x, m2, dim=2, log_softmax=True))
    x2 = torch.randn(4, 9, generator=g)
    mm = torch.tensor([[1] * 9, [1, 1, 1, 0, 0, 0, 0, 0, 0], [0, 1, 0, 1, 0, 1, 0, 1, 0], [0] * 9])
    out["x2"] = _np(x2)
    out["mask_last"] = _np(mm)
    out["y_last"] = _np(ref_att.masked_softmax(x2, mm))
    out["y_last_log"] = _np(ref_att.masked_softmax(x2, mm, log_softmax=True))
    _save("g1_masked_softmax", **out)


def _attn_case(ref_att, seed, B, T, M, D, text_mask, mod_mask):
    torch.manual_seed(seed)
    att = ref_att.BiDAFAttention(D, drop_prob=0.0)
    with torch.no_grad():
        att.bias.fill_(0.3)  # non-zero so the bias path is exercised
    att.eval()
    g = torch.Generator().manual_seed(seed + 1000)
    text = torch.randn(B, T, D, generator=g, requires_grad=True)
    mod = torch.randn(B, M, D, generator=g, requires_grad=True)
    cot = torch.randn(B, T, 4 * D, generator=g)
    s = att.get_similarity_matrix(text, mod)
    out = att(text, mod, text_mask, mod_mask)
    (out * cot).sum().backward()
    return dict(
        text=_np(text), mod=_np(mod), text_mask=_np(text_mask), mod_mask=_np(mod_mask), cot=_np(cot),
        w_t=_np(att.text_weight), w_m=_np(att.modality_weight), w_tm=_np(att.text_modality_weight),
        bias=_np(att.bias), sim=_np(s), out=_np(out), d_text=_np(text.grad), d_mod=_np(mod.grad),
        d_w_t=_np(att.text_weight.grad), d_w_m=_np(att.modality_weight.grad),
        d_w_tm=_np(att.text_modality_weight.grad), d_bias=_np(att.bias.grad))


def _prefix(lens, n):
    return torch.arange(n)[None, :] < torch.tensor(lens)[:, None]


def gen_attention(ref_att):
    """G2+G3: BiDAFAttention.get_similarity_matrix / forward / autograd (attention.py:37-75)."""
    cases = {}
    # G2-sized + full masks
    cases["small_full"] = _attn_case(ref_att, 2, 2, 7, 5, 8, _prefix([7, 7], 7), _prefix([5, 5], 5))
    cases["ragged"] = _attn_case(ref_att, 3, 3, 9, 6, 8, _prefix([9, 4, 1], 9), _prefix([6, 2, 3], 6))
    tm = torch.tensor([[1, 0, 1, 1, 0, 1, 1, 0], [0, 0, 0, 0, 0, 0, 0, 0]], dtype=torch.bool)
    mm = torch.tensor([[0, 1, 1, 0, 1], [0, 0, 0, 0, 0]], dtype=torch.bool)
    cases["nonprefix"] = _attn_case(ref_att, 4, 2, 8, 5, 12, tm, mm)
    cases["m1"] = _attn_case(ref_att, 5, 2, 6, 1, 8, _prefix([6, 3], 6), _prefix([1, 1], 1))
    cases["t1"] = _attn_case(ref_att, 6, 2, 1, 4, 8, _prefix([1, 1], 1), _prefix([4, 2], 4))
    # cfg-1 shapes, D = 2H = 200
    cases["cfg1_audio"] = _attn_case(ref_att, 7, 2, 50, 32, 200, _prefix([50, 17], 50), _prefix([32, 9], 32))
    cases["cfg1_image"] = _attn_case(ref_att, 8, 2, 50, 8, 200, _prefix([31, 50], 50), _prefix([8, 2], 8))
    flat = {}
    for cname, c in cases.items():
        for k, v in c.items():
            flat[cname + "__" + k] = v
    _save("g3_bidaf_attention", **flat)


def _rnn_case(ref_enc, seed, B, T, I, H, L, lengths):
    torch.manual_seed(seed)
    enc = ref_enc.RNNEncoder(I, H, L, drop_prob=0.0)
    enc.eval()
    g = torch.Generator().manual_seed(seed + 2000)
    x = torch.randn(B, T, I, generator=g, requires_grad=True)
    cot_y = torch.randn(B, T, 2 * H, generator=g)
    cot_h = torch.randn(B, 2 * L, H, generator=g)
    y, hn = enc(x, lengths)
    ((y * cot_y).sum() + (hn * cot_h).sum()).backward()
    out = dict(x=_np(x), lengths=np.asarray(lengths, dtype=np.int64), cot_y=_np(cot_y), cot_h=_np(cot_h),
               y=_np(y), h_n=_np(hn), d_x=_np(x.grad))
    for n, p in enc.named_parameters():
        out["param__" + n] = _np(p)
        out["grad__" + n] = _np(p.grad)
    return out


def gen_rnn(ref_enc):
    """G4: RNNEncoder forward / autograd (layers/encoding.py:62-108), incl. Q3 tie order."""
    cases = {}
    cases["l1_ragged"] = _rnn_case(ref_enc, 11, 5, 9, 6, 6, 1, [9, 1, 5, 7, 3])
    cases["l1_ties"] = _rnn_case(ref_enc, 12, 5, 8, 6, 6, 1, [5, 7, 5, 7, 5])
    cases["l2_i8h"] = _rnn_case(ref_enc, 13, 4, 7, 40, 5, 2, [7, 2, 7, 4])
    cases["l1_full"] = _rnn_case(ref_enc, 14, 3, 6, 4, 4, 1, [6, 6, 6])
    cases["l1_h100"] = _rnn_case(ref_enc, 15, 3, 12, 20, 100, 1, [12, 5, 9])
    cases["l2_h25"] = _rnn_case(ref_enc, 16, 3, 10, 200, 25, 2, [7, 10, 10])
    flat = {}
    for cname, c in cases.items():
        for k, v in c.items():
            flat[cname + "__" + k] = v
    _save("g4_rnn_encoder", **flat)


def gen_hot_region(ref_models):
    """G5: hook captures inside a real MMBiDAF (models.py:94-206) at cfg-1 lengths, drop_prob = 0.

    Hidden/embedding sizes are reduced (H=16) so the fixture stays small: the oracle is
    size-generic and the H=100 sizes are compared oracle<->HIP on the GPU."""
    B, T, Ma, Mi, H = 3, 50, 32, 8, 16
    Et, Ea, Ei = 24, 12, 20
    torch.manual_seed(224)
    model = ref_models.MMBiDAF(H, Et, Ea, Ei, torch.device("cpu"), drop_prob=0.0, max_transcript_length=60)
    g = torch.Generator().manual_seed(99)
    text = torch.randn(B, T, Et, generator=g)
    audio = torch.randn(B, Ma, Ea, generator=g)
    images = torch.randn(B, Mi, 3, 4, 4, generator=g)
    tl, al, il = [50, 31, 17], [32, 20, 9], [8, 5, 2]
    steps = 4
    targets = torch.randint(0, 17, (B, steps, 1), generator=g).float()
    caps = {}

    def hook(name):
        def fn(mod, inp, out):
            if isinstance(out, tuple):
                caps[name + "__y"] = _np(out[0])
                caps[name + "__h"] = _np(out[1])
            else:
                caps[name] = _np(out)
            if name in ("text_enc", "audio_enc", "image_enc"):
                caps[name + "__x"] = _np(inp[0])
        return fn

    hs = [getattr(model, n).register_forward_hook(hook(n)) for n in
          ("text_enc", "audio_enc", "image_enc", "bidaf_att_audio", "bidaf_att_image", "mod_t_a", "mod_t_i")]
    out = dict(text=_np(text), audio=_np(audio), images=_np(images), text_len=np.asarray(tl), audio_len=np.asarray(al),
               image_len=np.asarray(il), targets=_np(targets), resnet_w=_np(model.image_keyframes_emb.resnet.fc.weight),
               resnet_b=_np(model.image_keyframes_emb.resnet.fc.bias))
    model.train()
    dist, loss = model(text, tl, audio, al, images, il, targets, [steps] * B, steps)
    out["train_dist"] = _np(dist)
    out["train_loss"] = _np(loss)
    for k, v in caps.items():
        out["cap__" + k] = v
    model.zero_grad()
    loss.backward()
    for n, p in model.named_parameters():
        if p.grad is not None and not n.startswith("image_keyframes_emb"):
            out["grad__" + n] = _np(p.grad)
    model.eval()
    with torch.no_grad():
        dist_e, loss_e = model(text, tl, audio, al, images, il, targets, [steps] * B, steps)
    out["eval_dist"] = _np(dist_e)
    out["eval_loss"] = _np(loss_e)
    for h in hs:
        h.remove()
    for n, p in model.state_dict().items():
        if not n.startswith("image_keyframes_emb"):
            out["param__" + n] = _np(p)
    _save("g5_hot_region", **out)

    keys = [[n, list(p.shape)] for n, p in model.state_dict().items() if not n.startswith("image_keyframes_emb")]
    with open(os.path.join(OUT, "state_dict_keys.json"), "w") as f:
        json.dump(keys, f, indent=0)
    print("wrote state_dict_keys.json", len(keys), "tensors")


def _store_grads(out, named_params, prefix="grad__"):
    from golden_recipe import projections
    for n, p in named_params:
        if p.grad is None or n.startswith("image_keyframes_emb"):
            continue
        for kind, v in projections(n, p.grad).items():
            out[f"{prefix}{n}__{kind}"] = _np(v)


def gen_rnn_modelling_shape(ref_enc):
    """G7: the modelling-encoder shape of the model (models.py:70-78): RNNEncoder(8H=800, H=100, L=2), ragged lengths
    with ties (Q3).  Parameters by the shared recipe (tests/golden_recipe.py), big gradients as projections."""
    from golden_recipe import fill_parameters
    enc = ref_enc.RNNEncoder(800, 100, 2, drop_prob=0.0)
    enc.eval()
    csum = fill_parameters(list(enc.named_parameters()), seed=800)
    g = torch.Generator().manual_seed(801)
    B, T = 4, 16
    lengths = [16, 9, 16, 9]
    x = (torch.randn(B, T, 800, generator=g) * 0.5).requires_grad_(True)
    cot_y = torch.randn(B, T, 200, generator=g)
    cot_h = torch.randn(B, 4, 100, generator=g)
    y, hn = enc(x, lengths)
    ((y * cot_y).sum() + (hn * cot_h).sum()).backward()
    out = dict(x=_np(x), lengths=np.asarray(lengths, dtype=np.int64), cot_y=_np(cot_y), cot_h=_np(cot_h), y=_np(y), h_n=_np(hn),
               d_x=_np(x.grad), param_checksum=np.asarray(csum, dtype=np.float64))
    _store_grads(out, list(enc.named_parameters()))
    _save("g7_modelling_encoder_h100", **out)


def gen_model_h100(ref_models):
    """G8: a real reference MMBiDAF at cfg-1 lengths with the MODEL's hidden size H=100 (G5 uses H=16): hot-path
    captures, output distributions, loss and gradients (projections), train mode, drop_prob = 0."""
    from golden_recipe import fill_parameters
    B, T, Ma, Mi, H = 3, 50, 32, 8, 100
    Et, Ea, Ei = 24, 12, 20
    model = ref_models.MMBiDAF(H, Et, Ea, Ei, torch.device("cpu"), drop_prob=0.0, max_transcript_length=60)
    csum = fill_parameters(list(model.named_parameters()), seed=100)
    g = torch.Generator().manual_seed(101)
    text = torch.randn(B, T, Et, generator=g)
    audio = torch.randn(B, Ma, Ea, generator=g)
    images = torch.randn(B, Mi, 3, 4, 4, generator=g)
    tl, al, il = [50, 31, 17], [32, 20, 9], [8, 5, 2]
    steps = 4
    targets = torch.randint(0, 17, (B, steps, 1), generator=g).float()
    caps = {}

    def hook(name):
        def fn(mod, inp, out):
            if isinstance(out, tuple):
                caps[name + "__y"] = _np(out[0])
                caps[name + "__h"] = _np(out[1])
            else:
                caps[name] = _np(out[1:2])          # the ragged middle sample only (keeps the fixture small)
            if name in ("text_enc", "audio_enc", "image_enc"):
                caps[name + "__x"] = _np(inp[0])
        return fn

    hs = [getattr(model, n).register_forward_hook(hook(n)) for n in
          ("text_enc", "audio_enc", "image_enc", "bidaf_att_audio", "bidaf_att_image", "mod_t_a", "mod_t_i")]
    out = dict(text=_np(text), audio=_np(audio), images=_np(images), text_len=np.asarray(tl), audio_len=np.asarray(al),
               image_len=np.asarray(il), targets=_np(targets), resnet_w=_np(model.image_keyframes_emb.resnet.fc.weight),
               resnet_b=_np(model.image_keyframes_emb.resnet.fc.bias), param_checksum=np.asarray(csum, dtype=np.float64))
    model.train()
    dist, loss = model(text, tl, audio, al, images, il, targets, [steps] * B, steps)
    out["train_dist"] = _np(dist)
    out["train_loss"] = _np(loss)
    for k, v in caps.items():
        out["cap__" + k] = v
    model.zero_grad()
    loss.backward()
    _store_grads(out, list(model.named_parameters()))
    for h in hs:
        h.remove()
    _save("g8_model_h100", **out)


def gen_training_trajectory(ref_models, ref_util):
    """G9: THREE optimiser steps of the reference exactly as its training driver takes them (train.py:91-157): the model wrapped
    in nn.DataParallel with the device ids util.get_available_devices() reports (train.py:42,92 -- none on this CPU: the wrapper
    then calls the module directly), model.train(), util.EMA(model, 0.999) (train.py:100, args.py:47), Adadelta(lr 0.5, weight
    decay 0) (train.py:110, args.py:17-24), constant LambdaLR (train.py:111); per step: zero_grad, forward, loss.item(),
    backward, clip_grad_norm_(2.0) (train.py:154, args.py:41), optimizer.step(), scheduler.step(step // batch_size),
    ema(model, step // batch_size), step += batch_size (train.py:141-160).  cfg-1 shapes, the model's H = 100, drop_prob 0 (the one
    thing that is not train.py:210's: dropout draws cannot be replayed across devices), parameters by the shared recipe.
    Stored: the inputs, the loss of every step, and after every step the projections (tests/golden_recipe.py) of every
    parameter; after the last step also those of the EMA shadow and the total gradient norms clip_grad_norm_ returned."""
    import torch.optim as optim
    import torch.optim.lr_scheduler as sched
    from golden_recipe import fill_parameters, projections
    B, T, Ma, Mi, H = 3, 50, 32, 8, 100
    Et, Ea, Ei = 24, 12, 20
    device, gpu_ids = ref_util.get_available_devices()
    model = ref_models.MMBiDAF(H, Et, Ea, Ei, device, 0.0, 60)
    csum = fill_parameters(list(model.named_parameters()), seed=900, bound=0.3)
    model = nn.DataParallel(model, gpu_ids)
    model = model.to(device)
    model.train()
    ema = ref_util.EMA(model, 0.999)
    optimizer = optim.Adadelta(model.parameters(), 0.5, weight_decay=0)
    scheduler = sched.LambdaLR(optimizer, lambda s: 1.)
    g = torch.Generator().manual_seed(901)
    n_steps, dec_steps = 3, 9
    out = dict(param_checksum=np.asarray(csum, dtype=np.float64), n_steps=np.asarray(n_steps),
               resnet_w=_np(model.module.image_keyframes_emb.resnet.fc.weight), resnet_b=_np(model.module.image_keyframes_emb.resnet.fc.bias))
    lens = [([50, 31, 17], [32, 20, 9], [8, 5, 2]), ([44, 50, 9], [32, 7, 25], [3, 8, 8]), ([50, 50, 26], [11, 32, 32], [8, 1, 6])]
    step = 0
    losses, norms = [], []
    for k in range(n_steps):
        text = torch.randn(B, T, Et, generator=g)
        audio = torch.randn(B, Ma, Ea, generator=g)
        images = torch.randn(B, Mi, 3, 4, 4, generator=g)
        targets = torch.randint(0, 9, (B, dec_steps, 1), generator=g).float()
        tl, al, il = lens[k]
        out.update({f"s{k}__text": _np(text), f"s{k}__audio": _np(audio), f"s{k}__images": _np(images), f"s{k}__targets": _np(targets),
                    f"s{k}__text_len": np.asarray(tl), f"s{k}__audio_len": np.asarray(al), f"s{k}__image_len": np.asarray(il)})
        original_target_len = torch.tensor([dec_steps] * B)
        max_dec_len = torch.max(original_target_len)
        batch_size = text.size(0)
        optimizer.zero_grad()
        _, loss = model(text, tl, audio, al, images, il, targets, original_target_len, max_dec_len)
        losses.append(loss.item())
        loss.backward()
        norms.append(float(nn.utils.clip_grad_norm_(model.parameters(), 2.0)))
        optimizer.step()
        scheduler.step(step // batch_size)
        ema(model, step // batch_size)
        step += batch_size
        for n, p in model.named_parameters():
            if p.requires_grad and "image_keyframes_emb" not in n:
                for kind, v in projections(n, p.data).items():
                    out[f"s{k}__param__{n}__{kind}"] = _np(v)
    for n, v in ema.shadow.items():
        if "image_keyframes_emb" not in n:
            for kind, pv in projections(n, v).items():
                out[f"ema__{n}__{kind}"] = _np(pv)
    out["losses"] = np.asarray(losses, dtype=np.float64)
    out["grad_norms"] = np.asarray(norms, dtype=np.float64)
    _save("g9_training_trajectory", **out)


def main():
    sys.path.insert(0, TESTS_DIR)
    os.makedirs(OUT, exist_ok=True)
    _install_torchvision_stub()
    import json as _json
    sys.modules.setdefault("ujson", _json)     # util.py:13 imports ujson (absent here; used by nothing G9 touches): the stdlib module stands in
    sys.path.insert(0, REF)
    import layers.attention as ref_att
    import layers.encoding as ref_enc
    import models as ref_models
    import util as ref_util
    torch.set_num_threads(1)  # fixed summation order on the generating side
    only = sys.argv[1:]
    if not only or "g1-g8" in only:
        gen_masked_softmax(ref_att)
        gen_attention(ref_att)
        gen_rnn(ref_enc)
        gen_hot_region(ref_models)
        gen_rnn_modelling_shape(ref_enc)
        gen_model_h100(ref_models)
    if not only or "g9" in only:
        gen_training_trajectory(ref_models, ref_util)


if __name__ == "__main__":
    main()
