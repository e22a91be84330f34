"""ORACLE -- TEST INFRASTRUCTURE ONLY.  CPU (torch fp32) restatement of the MMBiDAF hot path.

Nothing in the product package (`mmbidaf_amd/`, `layers/`, `models.py`) imports this file.
Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may use it,
and only as the checker / reported baseline -- never as the thing measured or shipped.

Parity status: PINNED.  Every function below is checked in `tests/test_oracle_golden.py`
against fixtures under `tests/golden/` that were produced by importing and running the real
reference in the build container (`tools/gen_goldens.py`; reference = amankhullar/MMBiDAF,
arithmetic delegated by the reference to third-party `torch`, here 2.10.0 CPU).

What is restated (reference file:line):
  masked_softmax            layers/attention.py:78-98
  similarity_matrix         layers/attention.py:56-75   (BiDAFAttention.get_similarity_matrix)
  bidaf_attention           layers/attention.py:37-54   (BiDAFAttention.forward)
  lstm_packed_bidir         torch.nn.LSTM as called at layers/encoding.py:79-81,96 on a packed,
                            length-sorted batch (published LSTM equations, gate order i,f,g,o)
  rnn_encoder               layers/encoding.py:83-108   (RNNEncoder.forward, incl. sorted h_n)
  rnn_encoder_aten          same, but through torch's own packed nn.LSTM kernels exactly as the
                            reference call site does (used as the fast CPU baseline)
  get_mask                  models.py:86-92
  hot_region                models.py:97,102,113,116-118,131-135,143
The *_manual functions restate the analytic backward passes that the HIP kernels implement;
they are themselves checked against torch autograd of the forward restatement.
"""
import math

import torch
import torch.nn.functional as F

NEG = -1e30  # attention.py:94


# ----------------------------------------------------------------------------- attention
def masked_softmax(logits, mask, dim=-1, log_softmax=False):
    """attention.py:78-98: softmax(mask*x + (1-mask)*-1e30) -- an additive blend (Q1)."""
    m = mask.to(torch.float32)
    blended = m * logits + (1.0 - m) * NEG
    return F.log_softmax(blended, dim) if log_softmax else F.softmax(blended, dim)


def similarity_matrix(text, mod, w_t, w_m, w_tm, bias):
    """attention.py:56-75 (eval mode / already-dropped inputs).

    text (B,T,D), mod (B,M,D), w_t (D,1), w_m (D,1), w_tm (1,1,D), bias (1,) -> (B,T,M)."""
    T, M = text.size(1), mod.size(1)
    s0 = torch.matmul(text, w_t).expand(-1, -1, M)
    s1 = torch.matmul(mod, w_m).transpose(1, 2).expand(-1, T, -1)
    s2 = torch.matmul(text * w_tm, mod.transpose(1, 2))
    return s0 + s1 + s2 + bias


def bidaf_attention(text, mod, text_mask, mod_mask, w_t, w_m, w_tm, bias, text_d=None, mod_d=None):
    """attention.py:37-54.  text_d / mod_d are the dropped copies seen only by the similarity
    (attention.py:66-67, Q6); None = eval mode."""
    B, T, _ = text.shape
    M = mod.size(1)
    s = similarity_matrix(text if text_d is None else text_d, mod if mod_d is None else mod_d,
                          w_t, w_m, w_tm, bias)
    s1 = masked_softmax(s, mod_mask.view(B, 1, M), dim=2)
    s2 = masked_softmax(s, text_mask.view(B, T, 1), dim=1)
    a = torch.bmm(s1, mod)
    b = torch.bmm(torch.bmm(s1, s2.transpose(1, 2)), text)
    return torch.cat([text, a, text * a, text * b], dim=2)


def _softmax_stats(s_masked, dim):
    mx = s_masked.max(dim=dim, keepdim=True).values
    e = torch.exp(s_masked - mx)
    return e / e.sum(dim=dim, keepdim=True)


def bidaf_attention_manual(text, mod, text_mask, mod_mask, w_t, w_m, w_tm, bias, d_out,
                           text_d=None, mod_d=None):
    """Forward with the re-association b = s1 . (s2^T . text) and the analytic backward that the
    HIP kernels implement.  Returns (out, grads dict).  fp32 torch, no autograd."""
    B, T, D = text.shape
    M = mod.size(1)
    td = text if text_d is None else text_d
    md = mod if mod_d is None else mod_d
    wt, wm, wtm = w_t.view(D), w_m.view(D), w_tm.view(D)
    r = td @ wt + bias                      # (B,T)   row term
    c = md @ wm                             # (B,M)   column term
    S = torch.bmm(td * wtm, md.transpose(1, 2)) + r[:, :, None] + c[:, None, :]
    m1 = mod_mask.view(B, 1, M).to(torch.float32)
    m2 = text_mask.view(B, T, 1).to(torch.float32)
    neg = torch.full_like(S, NEG)
    P1 = _softmax_stats(torch.where(m1 > 0, S, neg), 2)     # row softmax over modality
    P2 = _softmax_stats(torch.where(m2 > 0, S, neg), 1)     # column softmax over text
    a = torch.bmm(P1, mod)
    q = torch.bmm(P2.transpose(1, 2), text)                 # (B,M,D)
    b = torch.bmm(P1, q)
    out = torch.cat([text, a, text * a, text * b], dim=2)

    g0, g1, g2, g3 = d_out.split(D, dim=2)
    da = g1 + g2 * text
    db = g3 * text
    d_text = g0 + g2 * a + g3 * b
    dP1 = torch.bmm(da, mod.transpose(1, 2)) + torch.bmm(db, q.transpose(1, 2))
    delta1 = (da * a).sum(2) + (db * b).sum(2)              # = sum_j P1 dP1
    dS1 = P1 * (dP1 - delta1[:, :, None]) * m1
    d_mod = torch.bmm(P1.transpose(1, 2), da)
    dq = torch.bmm(P1.transpose(1, 2), db)                  # (B,M,D)
    dP2 = torch.bmm(text, dq.transpose(1, 2))
    delta2 = (q * dq).sum(2)                                # (B,M) = sum_i P2 dP2
    dS2 = P2 * (dP2 - delta2[:, None, :]) * m2
    d_text = d_text + torch.bmm(P2, dq)
    dS = dS1 + dS2
    dr = dS.sum(2)
    dc = dS.sum(1)
    dX = torch.bmm(dS, md)                                  # wrt (td * wtm)
    dY = torch.bmm(dS.transpose(1, 2), td * wtm)            # wrt md
    d_td = dr[:, :, None] * wt + dX * wtm
    d_md = dc[:, :, None] * wm + dY
    grads = dict(
        d_w_t=(dr[:, :, None] * td).sum((0, 1)).view(D, 1),
        d_w_m=(dc[:, :, None] * md).sum((0, 1)).view(D, 1),
        d_w_tm=(dX * td).sum((0, 1)).view(1, 1, D),
        d_bias=dr.sum().view(1))
    if text_d is None:
        grads["d_text"] = d_text + d_td
    else:
        grads["d_text"], grads["d_text_d"] = d_text, d_td
    if mod_d is None:
        grads["d_mod"] = d_mod + d_md
    else:
        grads["d_mod"], grads["d_mod_d"] = d_mod, d_md
    return out, grads


# ----------------------------------------------------------------------------- LSTM
def lstm_layer_dir(x, lengths, w_ih, w_hh, b_ih, b_hh, reverse):
    """One direction of one layer of a packed LSTM: published equations
    (i,f,g,o = split(W_ih x_t + b_ih + W_hh h_{t-1} + b_hh); c = f*c + i*g; h = o*tanh(c)),
    with packed-sequence semantics: sample b runs exactly lengths[b] steps (reverse: from
    t = len-1 down to 0), padded outputs are 0, (h_n, c_n) are the state after its last step.
    x (B,T,I) -> y (B,T,H), h_n (B,H), c_n (B,H)."""
    B, T, _ = x.shape
    H = w_hh.size(1)
    gx = x @ w_ih.t() + (b_ih + b_hh)
    h = x.new_zeros(B, H)
    c = x.new_zeros(B, H)
    ys = [None] * T
    lens = torch.as_tensor(lengths, dtype=torch.long)
    order = range(T - 1, -1, -1) if reverse else range(T)
    for t in order:
        g = gx[:, t] + h @ w_hh.t()
        i, f, gg, o = g.split(H, dim=1)
        c_new = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        h_new = torch.sigmoid(o) * torch.tanh(c_new)
        live = (t < lens).to(x.dtype)[:, None]
        c = live * c_new + (1 - live) * c
        h = live * h_new + (1 - live) * h
        ys[t] = live * h_new
    return torch.stack(ys, dim=1), h, c


def lstm_packed_bidir(x, lengths, params, num_layers, dropout_masks=None):
    """Bidirectional multi-layer LSTM (nn.LSTM(batch_first, bidirectional) on a packed batch).
    params: dict with torch's names weight_ih_l{k}[_reverse] ... (encoding.py:79-81).
    dropout_masks: optional list (len L-1) of (B,T,2H) multiplicative masks applied between
    layers (nn.LSTM inter-layer dropout, training mode).  Returns y (B,T,2H), h_n (2L,B,H)."""
    inp = x
    hs = []
    for k in range(num_layers):
        outs = []
        for suffix, rev in (("", False), ("_reverse", True)):
            y, h, _ = lstm_layer_dir(inp, lengths, params[f"weight_ih_l{k}{suffix}"],
                                     params[f"weight_hh_l{k}{suffix}"], params[f"bias_ih_l{k}{suffix}"],
                                     params[f"bias_hh_l{k}{suffix}"], rev)
            outs.append(y)
            hs.append(h)
        inp = torch.cat(outs, dim=2)
        if dropout_masks is not None and k < num_layers - 1:
            inp = inp * dropout_masks[k]
    return inp, torch.stack(hs, dim=0)


def sort_lengths(lengths):
    """encoding.py:85,91: float-cast lengths, torch.sort(descending=True) (tie order is torch's, Q3)."""
    l = torch.Tensor(lengths)
    return l.sort(0, descending=True)


def rnn_encoder(x, lengths, params, num_layers, out_mask=None, dropout_masks=None):
    """encoding.py:83-108: y in original batch order, h_n (B,2L,H) in LENGTH-SORTED order (Q3)."""
    _, sort_idx = sort_lengths(lengths)
    y, h_n = lstm_packed_bidir(x, lengths, params, num_layers, dropout_masks)
    if out_mask is not None:               # F.dropout on y, encoding.py:104
        y = y * out_mask
    return y, h_n[:, sort_idx].transpose(0, 1)


def rnn_encoder_aten(x, lengths, rnn, drop_prob=0.0, training=False):
    """The same op through torch's packed nn.LSTM, exactly as the reference call site does
    (encoding.py:85-106).  `rnn` is an nn.LSTM(batch_first=True, bidirectional=True)."""
    from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence
    lens = torch.Tensor(lengths)
    orig_len = x.size(1)
    lens, sort_idx = lens.sort(0, descending=True)
    packed = pack_padded_sequence(x[sort_idx], lens, batch_first=True)
    out, (h_n, _) = rnn(packed)
    out, _ = pad_packed_sequence(out, batch_first=True, total_length=orig_len)
    _, unsort_idx = sort_idx.sort(0)
    out = F.dropout(out[unsort_idx], drop_prob, training)
    return out, h_n.transpose(0, 1)


def lstm_layer_dir_manual(x, lengths, w_ih, w_hh, b_ih, b_hh, reverse, d_y, d_hn):
    """Forward + analytic BPTT of one layer-direction, as the HIP kernels implement it.
    d_y (B,T,H) cotangent of y, d_hn (B,H) cotangent of the final hidden state (or None).
    Returns y, h_n, dict(d_x, d_w_ih, d_w_hh, d_b)  (d_b_ih == d_b_hh == d_b)."""
    B, T, I = x.shape
    H = w_hh.size(1)
    lens = torch.as_tensor(lengths, dtype=torch.long)
    gx = x @ w_ih.t() + (b_ih + b_hh)
    h = x.new_zeros(B, H)
    c = x.new_zeros(B, H)
    y = x.new_zeros(B, T, H)
    gates = x.new_zeros(B, T, 4 * H)       # post-activation i,f,g,o
    cs = x.new_zeros(B, T, H)              # c_t
    hprev = x.new_zeros(B, T, H)           # h_{t-1} as seen by step t
    cprev = x.new_zeros(B, T, H)
    order = list(range(T - 1, -1, -1) if reverse else range(T))
    for t in order:
        live = (t < lens)
        g = gx[:, t] + h @ w_hh.t()
        i, f, gg, o = g.split(H, dim=1)
        i, f, gg, o = torch.sigmoid(i), torch.sigmoid(f), torch.tanh(gg), torch.sigmoid(o)
        c_new = f * c + i * gg
        h_new = o * torch.tanh(c_new)
        hprev[:, t], cprev[:, t] = h, c
        gates[:, t] = torch.cat([i, f, gg, o], 1)
        cs[:, t] = c_new
        lv = live.to(x.dtype)[:, None]
        c = lv * c_new + (1 - lv) * c
        h = lv * h_new + (1 - lv) * h
        y[:, t] = lv * h_new
    h_n = h
    # ---- BPTT
    dh = x.new_zeros(B, H) if d_hn is None else d_hn.clone()
    dc = x.new_zeros(B, H)
    da = x.new_zeros(B, T, 4 * H)          # pre-activation gate gradients
    for t in reversed(order):
        lv = (t < lens).to(x.dtype)[:, None]
        i, f, gg, o = gates[:, t].split(H, dim=1)
        tc = torch.tanh(cs[:, t])
        dh_t = dh + d_y[:, t]
        do = dh_t * tc
        dc_t = dc + dh_t * o * (1 - tc * tc)
        dai = dc_t * gg * i * (1 - i)
        daf = dc_t * cprev[:, t] * f * (1 - f)
        dag = dc_t * i * (1 - gg * gg)
        dao = do * o * (1 - o)
        da_t = torch.cat([dai, daf, dag, dao], 1) * lv
        da[:, t] = da_t
        # dead steps (t >= len) pass the running cotangents through untouched
        dh = lv * (da_t @ w_hh) + (1 - lv) * dh
        dc = lv * (dc_t * f) + (1 - lv) * dc
    d_x = da @ w_ih
    d_w_ih = da.reshape(B * T, 4 * H).t() @ x.reshape(B * T, I)
    d_w_hh = da.reshape(B * T, 4 * H).t() @ hprev.reshape(B * T, H)
    d_b = da.sum((0, 1))
    return y, h_n, dict(d_x=d_x, d_w_ih=d_w_ih, d_w_hh=d_w_hh, d_b=d_b)


# ----------------------------------------------------------------------------- model segment
def get_mask(max_len, lengths):
    """models.py:86-92: bool prefix mask (B, max_len)."""
    return torch.arange(max_len)[None, :] < torch.as_tensor(lengths, dtype=torch.long)[:, None]


def hot_region(x_text, x_aud, x_img, text_len, aud_len, img_len, P, masks=None):
    """models.py:97,102,113,116-118,131-135,143 on post-embedding features; drop_prob = 0 unless `masks` is given.
    masks (training mode, every entry a multiplicative keep/(1-p) mask of its tensor's shape): 'out_text', 'out_aud',
    'out_img' (RNNEncoder output dropout, encoding.py:104), 'att_a_text', 'att_a_mod', 'att_i_text', 'att_i_mod' (the dropped
    copies only the similarity sees, attention.py:66-67), 'inter_a', 'inter_i' (nn.LSTM inter-layer dropout of the two-layer
    modelling encoders, encoding.py:81), 'out_a', 'out_i' (their output dropout).
    P: dict of parameter dicts: 'text_enc','audio_enc','image_enc','mod_t_a','mod_t_i' (LSTM
    params by torch name) and 'bidaf_att_audio','bidaf_att_image' (text_weight, modality_weight,
    text_modality_weight, bias).  Returns dict of every intermediate the goldens capture."""
    out = {}
    M = masks or {}
    te, _ = rnn_encoder(x_text, text_len, P["text_enc"], 1, out_mask=M.get("out_text"))
    ae, _ = rnn_encoder(x_aud, aud_len, P["audio_enc"], 1, out_mask=M.get("out_aud"))
    ie, _ = rnn_encoder(x_img, img_len, P["image_enc"], 1, out_mask=M.get("out_img"))
    tm = get_mask(x_text.size(1), text_len)
    am = get_mask(x_aud.size(1), aud_len)
    im = get_mask(x_img.size(1), img_len)

    def att(name, mod, mask, mt, mm):
        p = P[name]
        kw = dict(text_d=te * mt, mod_d=mod * mm) if mt is not None else {}
        return bidaf_attention(te, mod, tm, mask, p["text_weight"], p["modality_weight"],
                               p["text_modality_weight"], p["bias"], **kw)

    ta = att("bidaf_att_audio", ae, am, M.get("att_a_text"), M.get("att_a_mod"))
    ti = att("bidaf_att_image", ie, im, M.get("att_i_text"), M.get("att_i_mod"))
    ya, ha = rnn_encoder(ta, text_len, P["mod_t_a"], 2, out_mask=M.get("out_a"),
                         dropout_masks=[M["inter_a"]] if "inter_a" in M else None)
    yi, hi = rnn_encoder(ti, text_len, P["mod_t_i"], 2, out_mask=M.get("out_i"),
                         dropout_masks=[M["inter_i"]] if "inter_i" in M else None)
    out.update(text_enc=te, audio_enc=ae, image_enc=ie, att_audio=ta, att_image=ti,
               mod_t_a=ya, mod_t_a_h=ha, mod_t_i=yi, mod_t_i_h=hi,
               decoder_hidden=(ha.sum(1) + hi.sum(1)).unsqueeze(1))
    return out


class HotRegionCPU(torch.nn.Module):
    """The same region through torch's own CPU kernels, exactly as the reference's modules call them
    (nn.LSTM on packed sequences, bmm/softmax attention): the CPU baseline that bench.py times on the
    host cores.  Built from a hot-region state dict (keys `text_enc.rnn.weight_ih_l0`, ...)."""

    def __init__(self, state_dict, hidden_size):
        super().__init__()
        H = hidden_size
        mk = lambda i, l: torch.nn.LSTM(i, H, l, batch_first=True, bidirectional=True)
        self.rnns = torch.nn.ModuleDict(dict(text_enc=mk(H, 1), audio_enc=mk(H, 1), image_enc=mk(H, 1),
                                             mod_t_a=mk(8 * H, 2), mod_t_i=mk(8 * H, 2)))
        for name, rnn in self.rnns.items():
            rnn.load_state_dict({k[len(name) + 5:]: v.detach().cpu() for k, v in state_dict.items()
                                 if k.startswith(name + ".rnn.")})
        self.att = torch.nn.ParameterDict()
        for a in ("bidaf_att_audio", "bidaf_att_image"):
            for p in ("text_weight", "modality_weight", "text_modality_weight", "bias"):
                self.att[a + "__" + p] = torch.nn.Parameter(state_dict[a + "." + p].detach().cpu().clone())

    def _att(self, name, text, mod, tm, mm):
        g = lambda p: self.att[name + "__" + p]
        return bidaf_attention(text, mod, tm, mm, g("text_weight"), g("modality_weight"), g("text_modality_weight"), g("bias"))

    def forward(self, x_text, x_aud, x_img, text_len, aud_len, img_len):
        te, _ = rnn_encoder_aten(x_text, text_len, self.rnns["text_enc"])
        ae, _ = rnn_encoder_aten(x_aud, aud_len, self.rnns["audio_enc"])
        ie, _ = rnn_encoder_aten(x_img, img_len, self.rnns["image_enc"])
        tm, am, im = get_mask(x_text.size(1), text_len), get_mask(x_aud.size(1), aud_len), get_mask(x_img.size(1), img_len)
        ta = self._att("bidaf_att_audio", te, ae, tm, am)
        ti = self._att("bidaf_att_image", te, ie, tm, im)
        ya, ha = rnn_encoder_aten(ta, text_len, self.rnns["mod_t_a"])
        yi, hi = rnn_encoder_aten(ti, text_len, self.rnns["mod_t_i"])
        return ya, ha, yi, hi, (ha.sum(1) + hi.sum(1)).unsqueeze(1)

    def named_grads(self):
        """{hot-region parameter name: grad} for comparison with the HIP modules."""
        out = {}
        for name, rnn in self.rnns.items():
            for k, p in rnn.named_parameters():
                out[f"{name}.rnn.{k}"] = p.grad
        for k, p in self.att.items():
            out[k.replace("__", ".")] = p.grad
        return out


# ------------------------------------------------------------------------------------------ decoder
def decoder_step(P, sent_embed, hidden, cell, enc_a, enc_i, coverage, mask):
    """One step of MultimodalAttentionDecoder.forward, reference layers/attention.py:145-186, restated op by op.
    P: the decoder's state dict ('W1.weight', ..., 'lstm.weight_ih_l0', ..., 'out.bias').
    sent_embed (B,1,E), hidden (B,1,H), cell (1,B,H), enc_* (B,T,2H), coverage (B,T,1), mask (B,L) bool.
    Returns (dist (B,L), hidden (B,1,H), cell (1,B,H), att_cov (B,T,1), coverage (B,T,1))."""
    lin = lambda n, x: F.linear(x, P[n + ".weight"], P[n + ".bias"])
    e1 = lin("v1", torch.tanh(lin("W1", enc_a) + lin("W2", hidden) + lin("Wc1", coverage)))     # attention.py:147
    a1 = F.softmax(e1, dim=1)                                                                      # :148
    c1 = torch.sum(a1 * enc_a, dim=1)                                                              # :149-150
    e2 = lin("v2", torch.tanh(lin("W3", enc_i) + lin("W4", hidden) + lin("Wc2", coverage)))       # :153
    a2 = F.softmax(e2, dim=1)
    c2 = torch.sum(a2 * enc_i, dim=1)
    eb1 = lin("v_beta_1", torch.tanh(lin("W_beta_1", c1.unsqueeze(1)) + lin("W_beta_2", hidden)))  # :161
    eb2 = lin("v_beta_2", torch.tanh(lin("W_beta_3", c2.unsqueeze(1)) + lin("W_beta_4", hidden)))  # :162
    beta = F.softmax(torch.cat((eb1, eb2), dim=1), dim=1)                                          # :163-164
    c3 = torch.sum(torch.stack((c1, c2), dim=1) * beta, dim=1)                                     # :165-166
    att_cov = torch.bmm(torch.cat((a1, a2), dim=2), beta)                                          # :167
    coverage = coverage + att_cov                                                                  # :177
    x = torch.cat((c3.unsqueeze(1), sent_embed), dim=2)                                            # :179
    # one step of the single-layer nn.LSTM (:181), gate order i,f,g,o
    h, c = hidden.transpose(0, 1)[0], cell[0]
    g = F.linear(x[:, 0], P["lstm.weight_ih_l0"], P["lstm.bias_ih_l0"]) + F.linear(h, P["lstm.weight_hh_l0"], P["lstm.bias_hh_l0"])
    gi, gf, gg, go = g.chunk(4, dim=1)
    c_new = torch.sigmoid(gf) * c + torch.sigmoid(gi) * torch.tanh(gg)
    h_new = torch.sigmoid(go) * torch.tanh(c_new)
    dist = masked_softmax(lin("out", h_new), mask)                                                 # :184
    return dist, h_new.unsqueeze(1), c_new.unsqueeze(0), att_cov, coverage


def decoder_loop_train(P, enc_a, enc_i, h0, X, mask):
    """Teacher-forced loop of models.py:157-176 around decoder_step: X (S,B,E) are the decoder inputs of every step.
    Returns dists (S,B,L), att_cov (S,B,T), coverage after each step (S,B,T)."""
    B, T = enc_a.shape[:2]
    hidden, cell = h0.unsqueeze(1), torch.zeros(1, B, h0.shape[1])
    cov = torch.zeros(B, T, 1)
    dists, acs, covs = [], [], []
    for s in range(X.shape[0]):
        dist, hidden, cell, ac, cov = decoder_step(P, X[s].unsqueeze(1), hidden, cell, enc_a, enc_i, cov, mask)
        dists.append(dist)
        acs.append(ac[:, :, 0])
        covs.append(cov[:, :, 0])
    return torch.stack(dists), torch.stack(acs), torch.stack(covs)


# ------------------------------------------------------------------------------------------ embedding
def embedding(x, P, num_layers=2):
    """Embedding.forward with drop_prob = 0: bias-free projection, then the highway layers -- reference
    layers/encoding.py:25-30 and :52-59.  P: the module's state dict ('proj.weight', 'hwy.gates.k.*', 'hwy.transforms.k.*')."""
    h = F.linear(x, P["proj.weight"])                                                            # encoding.py:27
    for k in range(num_layers):
        g = torch.sigmoid(F.linear(h, P[f"hwy.gates.{k}.weight"], P[f"hwy.gates.{k}.bias"]))      # :55
        t = F.relu(F.linear(h, P[f"hwy.transforms.{k}.weight"], P[f"hwy.transforms.{k}.bias"]))  # :56
        h = g * t + (1 - g) * h                                                                   # :57
    return h
