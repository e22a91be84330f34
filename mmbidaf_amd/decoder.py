"""Fused decoder loop: the pointer decoder of the reference (layers/attention.py:100-186 driven by
models.py:157-199) with ONE HIP kernel per decode step (and one per backward step) instead of ~60 small launches.

* the loop-invariant memory projections W1.enc_a + b1 / W3.enc_i + b3 (attention.py:147,153) are hoisted out of the loop;
* `decoder_loop` runs all teacher-forced steps inside a single autograd Function; its backward walks the steps in
  reverse, accumulates the memory gradients in place and turns the stacked pre-activation gradients into the weight
  gradients with one GEMM per weight (K = steps * batch);
* `decoder_greedy` is the forward-only loop of evaluation mode (argmax feedback on the device).
"""
import ctypes

import torch

from . import _lib
from . import functional as MF

# order of the decoder module's tensors handed to the kernels (mmb_decoder_params), as (module attribute, tensor)
_PARAM_SOURCES = [("W2", "weight"), ("W2", "bias"), ("W4", "weight"), ("W4", "bias"),
                  ("Wc1", "weight"), ("Wc1", "bias"), ("v1", "weight"), ("v1", "bias"),
                  ("Wc2", "weight"), ("Wc2", "bias"), ("v2", "weight"), ("v2", "bias"),
                  ("W_beta_1", "weight"), ("W_beta_1", "bias"), ("W_beta_2", "weight"), ("W_beta_2", "bias"),
                  ("W_beta_3", "weight"), ("W_beta_3", "bias"), ("W_beta_4", "weight"), ("W_beta_4", "bias"),
                  ("v_beta_1", "weight"), ("v_beta_1", "bias"), ("v_beta_2", "weight"), ("v_beta_2", "bias"),
                  ("lstm", "weight_ih_l0"), ("lstm", "weight_hh_l0"), ("lstm", "bias_ih_l0"), ("lstm", "bias_hh_l0"),
                  ("out", "weight"), ("out", "bias")]


def decoder_tensors(dec):
    """The 30 parameter tensors of a MultimodalAttentionDecoder in mmb_decoder_params order."""
    return [getattr(getattr(dec, m), t) for m, t in _PARAM_SOURCES]


def _params_struct(ws, H, E, L):
    p = _lib.DecoderParams()
    keep = []
    by_name = {}
    for name, w in zip(_lib.DECODER_PTRS, ws):
        w = MF._f32c(w.detach())
        keep.append(w)
        by_name[name] = w
        setattr(p, name, w.data_ptr())
    # derived copies for the forward products (tiny, once per loop): see mmb_decoder_params
    n = by_name
    derived = {
        "WhT": torch.cat((n["W2"], n["W4"], n["Wb2"], n["Wb4"], n["W_hh"]), dim=0).t().contiguous(),
        "bh": torch.cat((n["b2"], n["b4"], n["bb2"], n["bb4"], n["b_hh"])).contiguous(),
        "Wb1T": n["Wb1"].t().contiguous(), "Wb3T": n["Wb3"].t().contiguous(),
        "W_ihcT": n["W_ih"][:, :2 * H].t().contiguous(), "W_outT": n["W_out"].t().contiguous(),
    }
    for name in _lib.DECODER_T_PTRS:
        keep.append(derived[name])
        setattr(p, name, derived[name].data_ptr())
    p.H, p.E, p.L = H, E, L
    return p, keep


def _ptr(t):
    return None if t is None else t.data_ptr()


def _step_fwd(lib, P, enc_a, enc_i, proj_a, proj_i, h, c, cov, xproj, mask, dist, h_out, c_out, att_cov, cov_out, saved, scratch):
    B, T = cov.shape
    rc = lib.mmb_decoder_step_fwd(ctypes.byref(P), _ptr(enc_a), _ptr(enc_i), _ptr(proj_a), _ptr(proj_i), _ptr(h), _ptr(c),
                                  _ptr(cov), _ptr(xproj), _ptr(mask), _ptr(dist), _ptr(h_out), _ptr(c_out), _ptr(att_cov),
                                  _ptr(cov_out), _ptr(saved), _ptr(scratch), B, T, cov.device.index, MF._stream())
    _lib.check(rc, "mmb_decoder_step_fwd")


class _DecoderLoopFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, enc_a, enc_i, proj_a, proj_i, h0, X, mask, *ws):
        lib = _lib.load()
        MF._require_gpu(enc_a, enc_i, proj_a, proj_i, h0, X, mask, *ws)
        enc_a, enc_i, proj_a, proj_i, h0, X = (MF._f32c(t) for t in (enc_a, enc_i, proj_a, proj_i, h0, X))
        S, B, E = X.shape
        T, H = enc_a.shape[1], h0.shape[1]
        L = ws[28].shape[0]
        mask = MF._mask_u8(mask, B, L)
        dev = enc_a.device
        P, keep = _params_struct(ws, H, E, L)
        new = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)
        hs, cs, covs = new(S + 1, B, H), new(S + 1, B, H), new(S + 1, B, T)
        hs[0].copy_(h0)
        cs[0].zero_()
        covs[0].zero_()
        dists, att_covs = new(S, B, L), new(S, B, T)
        saved = new(S, B, lib.mmb_decoder_saved_floats(T, H))
        # x part of the LSTM input product, hoisted: one GEMM over all steps  (W_ih = [context columns | x columns])
        w_ihx = keep[24][:, 2 * H:].contiguous()
        xproj = MF.gemm(X.reshape(S * B, E), w_ihx, bias=keep[26], tb=True).reshape(S, B, 4 * H)
        scratch = new(lib.mmb_decoder_scratch_floats(B, T, H))
        for s in range(S):
            _step_fwd(lib, P, enc_a, enc_i, proj_a, proj_i, hs[s], cs[s], covs[s], xproj[s], mask, dists[s], hs[s + 1], cs[s + 1],
                      att_covs[s], covs[s + 1], saved[s], scratch)
        ctx.save_for_backward(enc_a, enc_i, proj_a, proj_i, X, mask, hs, cs, covs, dists, saved, *keep[:30])
        ctx.shapes = [w.shape for w in ws]
        return dists, att_covs, covs[1:]

    @staticmethod
    def backward(ctx, d_dists, d_att_covs, d_covs):
        lib = _lib.load()
        enc_a, enc_i, proj_a, proj_i, X, mask, hs, cs, covs, dists, saved, *ws = ctx.saved_tensors
        S, B, E = X.shape
        T, H = enc_a.shape[1], hs.shape[2]
        H2, L = 2 * H, dists.shape[2]
        dev = enc_a.device
        P, keep = _params_struct(ws, H, E, L)
        zeros = lambda *s: torch.zeros(*s, device=dev, dtype=torch.float32)
        new = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)
        d_dists = None if d_dists is None else MF._f32c(d_dists)
        d_att_covs = None if d_att_covs is None else MF._f32c(d_att_covs)
        d_covs = None if d_covs is None else MF._f32c(d_covs)
        d_proj_a, d_enc_a, d_proj_i, d_enc_i = (zeros(B, T, H2) for _ in range(4))
        dl_out, dl_g = new(S, B, L), new(S, B, 4 * H)
        dl_b1, dl_b2, dl_ha, dl_hi = (new(S, B, H2) for _ in range(4))
        vec = zeros(B, lib.mmb_decoder_vec_acc_floats(H))
        scratch = new(lib.mmb_decoder_scratch_floats(B, T, H))
        d_h, d_c, d_cov = [zeros(B, H), new(B, H)], [zeros(B, H), new(B, H)], [zeros(B, T), new(B, T)]
        cur = 0
        for s in reversed(range(S)):
            d_cov_out = d_cov[cur] if d_covs is None else d_cov[cur] + d_covs[s]
            rc = lib.mmb_decoder_step_bwd(
                ctypes.byref(P), _ptr(enc_a), _ptr(enc_i), _ptr(proj_a), _ptr(proj_i), _ptr(hs[s]), _ptr(cs[s]), _ptr(covs[s]),
                _ptr(mask), _ptr(saved[s]), _ptr(dists[s]), _ptr(cs[s + 1]),
                _ptr(None if d_dists is None else d_dists[s]), _ptr(d_h[cur]), _ptr(d_c[cur]),
                _ptr(None if d_att_covs is None else d_att_covs[s]), _ptr(d_cov_out),
                _ptr(d_h[cur ^ 1]), _ptr(d_c[cur ^ 1]), _ptr(d_cov[cur ^ 1]),
                _ptr(d_proj_a), _ptr(d_enc_a), _ptr(d_proj_i), _ptr(d_enc_i),
                _ptr(dl_out[s]), _ptr(dl_g[s]), _ptr(dl_b1[s]), _ptr(dl_b2[s]), _ptr(dl_ha[s]), _ptr(dl_hi[s]), _ptr(vec), _ptr(scratch),
                B, T, dev.index, MF._stream())
            _lib.check(rc, "mmb_decoder_step_bwd")
            cur ^= 1
        # ---- weight gradients: one GEMM per weight over the stacked (step, sample) rows
        flat = lambda t: t.reshape(S * B, -1)
        h_prev, h_new = flat(hs[:S]), flat(hs[1:])
        sv = saved.reshape(S * B, -1)
        ctx_a, ctx_i = sv[:, 2 * T:2 * T + H2], sv[:, 2 * T + H2:2 * T + 2 * H2]
        beta = sv[:, 2 * T + 4 * H2 + 4 * H:2 * T + 4 * H2 + 4 * H + 2]
        inp = torch.cat((beta[:, 0:1] * ctx_a + beta[:, 1:2] * ctx_i, flat(X)), dim=1)     # [c3 ; x] of every step
        tg = lambda delta, act: MF.gemm(flat(delta), act.contiguous(), ta=True)           # delta^T . act
        V = vec.sum(0)
        d_X = MF.gemm(flat(dl_g), ws[24][:, 2 * H:].contiguous()).reshape(S, B, E)      # gradient of the decoder inputs
        g = [None] * 30
        g[0], g[1] = tg(dl_ha, h_prev), flat(dl_ha).sum(0)                                  # W2, b2
        g[2], g[3] = tg(dl_hi, h_prev), flat(dl_hi).sum(0)                                  # W4, b4
        g[4], g[5], g[6], g[7] = V[0:H2], g[1].clone(), V[H2:2 * H2], V[6 * H2:6 * H2 + 1]  # Wc1, its bias, v1, its bias
        g[8], g[9], g[10], g[11] = V[2 * H2:3 * H2], g[3].clone(), V[3 * H2:4 * H2], V[6 * H2 + 1:6 * H2 + 2]
        g[12], g[13] = tg(dl_b1, ctx_a), flat(dl_b1).sum(0)                                 # W_beta_1
        g[14], g[15] = tg(dl_b1, h_prev), g[13].clone()                                     # W_beta_2
        g[16], g[17] = tg(dl_b2, ctx_i), flat(dl_b2).sum(0)                                 # W_beta_3
        g[18], g[19] = tg(dl_b2, h_prev), g[17].clone()                                     # W_beta_4
        g[20], g[21] = V[4 * H2:5 * H2], V[6 * H2 + 2:6 * H2 + 3]                           # v_beta_1
        g[22], g[23] = V[5 * H2:6 * H2], V[6 * H2 + 3:6 * H2 + 4]                           # v_beta_2
        g[24], g[25] = tg(dl_g, inp), tg(dl_g, h_prev)                                      # lstm W_ih, W_hh
        g[26] = flat(dl_g).sum(0)                                                           # lstm biases: equal gradients,
        g[27] = g[26].clone()                                                               # distinct storage (in-place grad ops)
        g[28], g[29] = tg(dl_out, h_new), flat(dl_out).sum(0)                               # out
        # no two gradients may share storage (AccumulateGrad keeps what it is handed; in-place grad ops such as
        # clip_grad_norm_ would hit an aliased pair twice): shared values are cloned above, slices of `V` here
        g = [(t.clone() if t._base is not None else t).reshape(shape) for t, shape in zip(g, ctx.shapes)]
        return (d_enc_a, d_enc_i, d_proj_a, d_proj_i, d_h[cur], d_X, None, *g)


def decoder_loop(dec, enc_a, enc_i, h0, X, mask):
    """Teacher-forced decode: X (S,B,E) decoder inputs of every step, h0 (B,H), mask (B,L).
    Returns dists (S,B,L), att_cov (S,B,T), coverage after each step (S,B,T)."""
    proj_a = MF.linear(enc_a, dec.W1.weight, dec.W1.bias)     # loop-invariant (attention.py:147), library GEMM
    proj_i = MF.linear(enc_i, dec.W3.weight, dec.W3.bias)     # (attention.py:153)
    return _DecoderLoopFn.apply(enc_a, enc_i, proj_a, proj_i, h0, X, mask, *decoder_tensors(dec))


@torch.no_grad()
def decoder_greedy(dec, enc_a, enc_i, h0, embedded_text, mask, steps):
    """Evaluation-mode loop (models.py:186-193): the next input is the embedding of the arg-max sentence.
    Returns dists (S,B,L), att_cov of the last step (B,T), coverage after the last step (B,T)."""
    lib = _lib.load()
    MF._require_gpu(enc_a, enc_i, h0, embedded_text, mask)
    enc_a, enc_i, h0, emb = (MF._f32c(t) for t in (enc_a, enc_i, h0, embedded_text))
    proj_a = MF.linear(enc_a, dec.W1.weight, dec.W1.bias)
    proj_i = MF.linear(enc_i, dec.W3.weight, dec.W3.bias)
    B, T, H2 = enc_a.shape
    H, E, L = h0.shape[1], emb.shape[2], dec.out.weight.shape[0]
    mask = MF._mask_u8(mask, B, L)
    P, keep = _params_struct(decoder_tensors(dec), H, E, L)
    dev = enc_a.device
    new = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)
    h, c, cov = [h0.clone(), new(B, H)], [torch.zeros(B, H, device=dev), new(B, H)], [torch.zeros(B, T, device=dev), new(B, T)]
    x = torch.zeros(B, E, device=dev)
    dists, att_cov = new(steps, B, L), new(B, T)
    rows = torch.arange(B, device=dev)
    w_ihx, b_ih = dec.lstm.weight_ih_l0[:, 2 * H:], dec.lstm.bias_ih_l0
    scratch = new(lib.mmb_decoder_scratch_floats(B, T, H))
    cur = 0
    for s in range(steps):
        xproj = MF._f32c(torch.nn.functional.linear(x, w_ihx, b_ih))
        _step_fwd(lib, P, enc_a, enc_i, proj_a, proj_i, h[cur], c[cur], cov[cur], xproj, mask, dists[s], h[cur ^ 1], c[cur ^ 1],
                  att_cov, cov[cur ^ 1], None, scratch)
        x = emb[rows, dists[s].argmax(dim=1)]
        cur ^= 1
    return dists, att_cov, cov[cur]
