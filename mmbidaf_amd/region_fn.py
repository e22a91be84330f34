"""The hot-path region as ONE autograd node with a lean host side.

`MMBiDAF.hot_path` (reference models.py:97,102,113,116-118,131-135,143) issued module by module costs the host ~2.1 ms of
Python per fwd+bwd step at the metric configuration -- seven autograd Functions, ~120 `torch.empty` calls, one ctypes
descriptor array per call, stream bookkeeping per tensor -- as long as the GPU needs for the step (2.2 ms), so a caller that
cannot replay a captured graph (train.py feeds new lengths every batch, train.py:126-146) is host-bound.  Here the same C-ABI
calls are issued by one `torch.autograd.Function`:

  * two allocations per direction (an arena for everything the backward pass needs, one for scratch) instead of one per
    tensor; the library gets raw addresses computed from the arena's base, no tensor views are made for internal buffers;
  * the index vectors a step derives from its lengths (three int32 length vectors, three positions in the reference's
    descending-length order, Q3) travel in ONE pinned host-to-device copy, cached per lengths;
  * the backward pass keeps round 2's schedule (critical path on the caller's stream; operand-plane preparation and the
    weight-gradient phase of layer k on the side stream beside the recurrence of layer k - 1) but as straight-line code with
    one join at its end: no deferred-work queue, no per-tensor `record_stream`, no end-of-backward engine callback.

Results are those of the modular path bit for bit in eval mode, and in training mode under the same generator state (the
eleven dropout masks are drawn by the same `F.dropout` calls in the same order).  The modular path stays the general one:
this node is taken only for the exact module structure of the reference's region on a GPU, with the fused attention width
(D = 2H <= 208), plain leaf parameters and no module hooks -- anything else falls back (`eligible`).
"""
import ctypes
import os

import torch
import torch.nn.functional as F

from . import _lib
from . import functional as MF
from .encoding import sorted_order

_ENABLED = os.environ.get("MMB_REGION_FN", "1") != "0"
PREPARE, HAVE_XC, HAVE_WT = 4, 8, 16


def _al(n):
    return (n + 255) // 256 * 256


class _Layout:
    """Byte offsets of named buffers inside one arena."""

    def __init__(self):
        self.off = {}
        self.size = 0

    def add(self, name, nbytes):
        self.off[name] = self.size
        self.size += _al(max(int(nbytes), 4))
        return self.off[name]


class _Plan:
    """Everything about a step that depends on the sizes only: arena layouts and workspace sizes (library queries made once)."""

    def __init__(self, B, T, Ma, Mi, H, drop):
        lib = _lib.load()
        self.dims = (B, T, Ma, Mi, H)
        self.drop = drop
        D = 2 * H
        f = 4
        keep, scr = _Layout(), _Layout()
        # LSTM problems: (tag, T, I, has hn_pos-ordered h_n)
        self.lstm = [("et", T, H), ("ea", Ma, H), ("ei", Mi, H), ("a0", T, 8 * H), ("i0", T, 8 * H), ("a1", T, 2 * H), ("i1", T, 2 * H)]
        for tag, Tn, I in self.lstm:
            keep.add(tag + ".y", B * Tn * D * f)
            keep.add(tag + ".gates", B * Tn * 8 * H * f)
            keep.add(tag + ".cs", B * Tn * D * f)
            keep.add(tag + ".hn", 2 * B * H * f)
            keep.add(tag + ".absmax", lib.mmb_bilstm_absmax_floats(B, Tn, H) * f)
            scr.add(tag + ".gx", B * Tn * 8 * H * f)
            scr.add(tag + ".cn", 2 * B * H * f)
            scr.add(tag + ".ws", lib.mmb_bilstm_ws_bytes(B, Tn, I, H, 0))
        self.att = [("aa", Ma), ("ai", Mi)]
        self.att_saved, self.att_ws_b = {}, {}
        for tag, M in self.att:
            keep.add(tag + ".out", B * T * 4 * D * f)
            keep.add(tag + ".bsave", B * T * D * f)
            keep.add(tag + ".rterm", B * T * f)
            keep.add(tag + ".cterm", B * M * f)
            keep.add(tag + ".rstat", B * T * 2 * f)
            keep.add(tag + ".cstat", B * M * 2 * f)
            self.att_saved[tag] = int(lib.mmb_bidaf_saved_bytes(B, T, M, D, int(drop)))
            keep.add(tag + ".saved", self.att_saved[tag])
            scr.add(tag + ".ws", max(int(lib.mmb_bidaf_fwd_workspace_bytes(B, T, M, D)), 256))
            self.att_ws_b[tag] = int(lib.mmb_bidaf_bwd_workspace_bytes(B, T, M, D))
        keep.add("hid_a", B * 4 * H * f)
        keep.add("hid_i", B * 4 * H * f)
        keep.add("dec", B * H * f)
        self.keep, self.scr = keep, scr
        # backward arena
        bw = _Layout()
        for tag, Tn, I in self.lstm:
            bw.add(tag + ".d_a", B * Tn * 8 * H * f)
            bw.add(tag + ".d_w_cat", 8 * H * (I + 2 * H) * f)
            bw.add(tag + ".ws", lib.mmb_bilstm_ws_bytes(B, Tn, I, H, 1))
            bw.add(tag + ".d_x", B * Tn * I * f)
            bw.add(tag + ".d_y", B * Tn * D * f)             # zero cotangent / masked cotangent staging
        bw.add("d_h", 4 * B * 2 * H * f)
        for tag, M in self.att:
            bw.add(tag + ".ws", self.att_ws_b[tag])
            bw.add(tag + ".d_text", B * T * D * f)
            bw.add(tag + ".d_mod", B * M * D * f)
            if drop:
                bw.add(tag + ".d_text_d", B * T * D * f)
                bw.add(tag + ".d_mod_d", B * M * D * f)
        self.bw = bw
        # training mode: the eleven dropout masks of a step are ONE F.dropout draw over a flat vector of ones, cut in this order
        # (the call order of the modular path: encoders' output dropout, dropped copies of (text, audio) and (text, image),
        # inter-layer dropout of the two modelling encoders, their output dropout)
        self.mask_layout = mask_layout(B, T, Ma, Mi, H)
        self.mask_total = self.mask_layout[-1][2] + (self.mask_layout[-1][3] + 3) // 4 * 4


MASK_NAMES = ("out_et", "out_ea", "out_ei", "aa_t", "aa_m", "ai_t", "ai_m", "inter_a", "inter_i", "out_a", "out_i")


def mask_layout(B, T, Ma, Mi, H):
    """[(name, shape, offset, numel)] of the eleven dropout masks inside the flat draw of a training-mode step."""
    D = 2 * H
    shapes = [(B, T, D), (B, Ma, D), (B, Mi, D), (B, T, D), (B, Ma, D), (B, T, D), (B, Mi, D), (B, T, D), (B, T, D), (B, T, D), (B, T, D)]
    out, o = [], 0
    for name, sh in zip(MASK_NAMES, shapes):
        n = sh[0] * sh[1] * sh[2]
        out.append((name, sh, o, n))
        o += (n + 3) // 4 * 4
    return out


def draw_masks(B, T, Ma, Mi, H, p, dev):
    """The masks a training-mode step of these sizes draws from torch's generator in its current state: {name: (shape) tensor of
    0 / 1/(1-p)}.  Tests replay a step's masks with it (same generator state -> same masks) and hand them to the oracle."""
    lay = mask_layout(B, T, Ma, Mi, H)
    total = lay[-1][2] + (lay[-1][3] + 3) // 4 * 4
    flat = F.dropout(_ones_flat(total, dev), p, True)
    return {name: flat[o:o + n].view(sh) for name, sh, o, n in lay}, flat


_plans = {}


def _plan(B, T, Ma, Mi, H, drop):
    key = (B, T, Ma, Mi, H, bool(drop))
    p = _plans.get(key)
    if p is None:
        if len(_plans) > 32:
            _plans.clear()
        p = _plans[key] = _Plan(B, T, Ma, Mi, H, bool(drop))
    return p


# ---- per-lengths device metadata: [len_t | len_a | len_i | pos_t | pos_a | pos_i] int32, one pinned copy
_meta_cache = {}


def _meta(dev, lens3):
    key = (dev.index, tuple(lens3[0]), tuple(lens3[1]), tuple(lens3[2]))
    m = _meta_cache.pop(key, None)
    if m is None:
        parts = [torch.tensor(list(l), dtype=torch.int32) for l in lens3]
        for l in lens3:
            order = sorted_order(l)              # the reference's own call (float cast + torch.sort, tie order included)
            inv = torch.empty_like(order)
            inv[order] = torch.arange(order.numel())
            parts.append(inv.to(torch.int32))
        host = torch.cat(parts).pin_memory()
        m = host.to(dev, non_blocking=True)
        if len(_meta_cache) >= 256:
            _meta_cache.pop(next(iter(_meta_cache)))
    _meta_cache[key] = m
    return m


def param_list(R):
    """The 64 parameters of the region in the fixed order the node takes them."""
    ps = []
    for enc in (R.text_enc, R.audio_enc, R.image_enc):
        rnn = enc.rnn
        ps += [rnn.weight_ih_l0, rnn.weight_hh_l0, rnn.bias_ih_l0, rnn.bias_hh_l0,
               rnn.weight_ih_l0_reverse, rnn.weight_hh_l0_reverse, rnn.bias_ih_l0_reverse, rnn.bias_hh_l0_reverse]
    for att in (R.bidaf_att_audio, R.bidaf_att_image):
        ps += [att.text_weight, att.modality_weight, att.text_modality_weight, att.bias]
    for enc in (R.mod_t_a, R.mod_t_i):
        rnn = enc.rnn
        for l in (0, 1):
            for sfx in ("", "_reverse"):
                ps += [getattr(rnn, f"weight_ih_l{l}{sfx}"), getattr(rnn, f"weight_hh_l{l}{sfx}"),
                       getattr(rnn, f"bias_ih_l{l}{sfx}"), getattr(rnn, f"bias_hh_l{l}{sfx}")]
    return ps


# index of the first parameter of each LSTM problem / attention in param_list
_P_LSTM = {"et": 0, "ea": 8, "ei": 16, "a0": 32, "a1": 40, "i0": 48, "i1": 56}
_P_ATT = {"aa": 24, "ai": 28}


_last_params = [None]
_static_ok = {}      # id(region module) -> (parameter ids, device, hidden size) of the last successful static check


def eligible(R, xs, lens3):
    """May this step take the single-node path?  (Everything else takes the modular path of model.MMBiDAF.hot_path.)"""
    if not _ENABLED:
        return False
    import torch.nn.modules.module as _M
    if _M._global_forward_hooks or _M._global_forward_pre_hooks or _M._global_backward_hooks:
        return False
    mods = (R.text_enc, R.audio_enc, R.image_enc, R.bidaf_att_audio, R.bidaf_att_image, R.mod_t_a, R.mod_t_i)
    for m in mods:
        if m._forward_hooks or m._forward_pre_hooks or m._backward_hooks or m._backward_pre_hooks:
            return False
        rnn = getattr(m, "rnn", None)
        if rnn is not None and (rnn._forward_hooks or rnn._forward_pre_hooks or rnn._backward_hooks):
            return False
    x_text, x_aud, x_img = xs
    if not (x_text.is_cuda and x_text.dtype == torch.float32 and x_aud.dtype == torch.float32 and x_img.dtype == torch.float32):
        return False
    if x_text.dim() != 3 or x_aud.dim() != 3 or x_img.dim() != 3:
        return False
    B, T, H = x_text.shape
    if x_aud.shape[0] != B or x_img.shape[0] != B or x_aud.shape[2] != H or x_img.shape[2] != H:
        return False
    if 2 * H > _lib.ATT_MAX_D or H % 4 != 0:
        return False
    if MF.get_precision() != "fp32":
        return False
    for l, n in zip(lens3, (T, x_aud.shape[1], x_img.shape[1])):
        if len(l) != B or min(l) < 1 or max(l) > n:
            return False      # (the modular path raises the reference-style error)
    tr = mods[0].training
    for m in mods[1:]:
        if m.training != tr:
            return False
    ps = param_list(R)
    dev = x_text.device
    for p in ps:
        if p._backward_hooks:
            return False
    sig = (tuple(map(id, ps)), dev, H)
    _last_params[0] = (id(R), ps)
    if _static_ok.get(id(R)) == sig:
        return True
    # shapes, dtypes, layout: a property of the parameter OBJECTS, checked once per set of them
    for enc, L, I in ((R.text_enc, 1, H), (R.audio_enc, 1, H), (R.image_enc, 1, H), (R.mod_t_a, 2, 8 * H), (R.mod_t_i, 2, 8 * H)):
        rnn = enc.rnn
        if rnn.num_layers != L or rnn.hidden_size != H or rnn.input_size != I or not rnn.bidirectional:
            return False
    if R.bidaf_att_audio.text_weight.shape[0] != 2 * H or R.bidaf_att_image.text_weight.shape[0] != 2 * H:
        return False
    for p in ps:
        if p.device != dev or p.dtype != torch.float32 or not p.is_contiguous():
            return False
    if len(_static_ok) > 64:
        _static_ok.clear()
    _static_ok[id(R)] = sig
    return True


def _drop_conf(R):
    """None: no dropout (eval mode or drop_prob 0); a float: the one dropout probability of a training-mode step; False: per-site
    probabilities differ (modular path)"""
    tr = R.text_enc.training
    p_enc = [R.text_enc.drop_prob, R.audio_enc.drop_prob, R.image_enc.drop_prob]
    p_att = [R.bidaf_att_audio.drop_prob, R.bidaf_att_image.drop_prob]
    p_mod = [R.mod_t_a.drop_prob, R.mod_t_i.drop_prob]
    p_inter = [R.mod_t_a.rnn.dropout, R.mod_t_i.rnn.dropout]
    ps = p_enc + p_att + p_mod + p_inter
    if not tr or all(p == 0.0 for p in ps):
        return None
    if any(p != ps[0] for p in ps) or not (0.0 < ps[0] < 1.0):
        return False            # different probabilities per site (the reference passes ONE drop_prob everywhere): modular path
    return ps[0]


_ones = {}


def _ones_flat(total, dev):
    key = (dev.index, total)
    o = _ones.get(key)
    if o is None:
        if len(_ones) > 8:
            _ones.clear()
        o = _ones[key] = torch.ones(total, device=dev, dtype=torch.float32)
    return o


class _Ctx:
    """What forward hands to backward besides tensors."""


class _RegionFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, st, x_text, x_aud, x_img, *params):
        lib = _lib.load()
        plan, meta, drop = st.plan, st.meta, st.drop
        B, T, Ma, Mi, H = plan.dims
        D = 2 * H
        dev = x_text.device
        di = dev.index
        stream = torch.cuda.current_stream(dev).cuda_stream
        xs = {"et": x_text.contiguous(), "ea": x_aud.contiguous(), "ei": x_img.contiguous()}
        keep = torch.empty(plan.keep.size, device=dev, dtype=torch.uint8)
        scr = torch.empty(plan.scr.size, device=dev, dtype=torch.uint8)
        kb, sb = keep.data_ptr(), scr.data_ptr()
        ko, so = plan.keep.off, plan.scr.off
        mp = meta.data_ptr()
        len_ptr = {"et": mp, "ea": mp + 4 * B, "ei": mp + 8 * B, "a0": mp, "a1": mp, "i0": mp, "i1": mp}
        pos_ptr = {"et": mp + 12 * B, "ea": mp + 16 * B, "ei": mp + 20 * B, "a0": mp + 12 * B, "a1": mp + 12 * B, "i0": mp + 12 * B, "i1": mp + 12 * B}
        Tn = {"et": T, "ea": Ma, "ei": Mi, "a0": T, "a1": T, "i0": T, "i1": T}
        In = {"et": H, "ea": H, "ei": H, "a0": 8 * H, "a1": 2 * H, "i0": 8 * H, "i1": 2 * H}
        pp = [p.data_ptr() for p in params]

        def view(off, shape):
            n = 1
            for s_ in shape:
                n *= s_
            return keep[off:off + 4 * n].view(torch.float32).view(shape)

        def lstm_fwd(tags, x_ptrs):
            n = len(tags)
            descs = (_lib.LstmFwdDesc * n)()
            for d, tag, xp in zip(descs, tags, x_ptrs):
                q = _P_LSTM[tag]
                d.x, d.lengths = xp, len_ptr[tag]
                d.w_ih[0], d.w_hh[0], d.b_ih[0], d.b_hh[0] = pp[q], pp[q + 1], pp[q + 2], pp[q + 3]
                d.w_ih[1], d.w_hh[1], d.b_ih[1], d.b_hh[1] = pp[q + 4], pp[q + 5], pp[q + 6], pp[q + 7]
                d.y, d.h_n, d.c_n = kb + ko[tag + ".y"], kb + ko[tag + ".hn"], sb + so[tag + ".cn"]
                d.gx, d.gates, d.cs = sb + so[tag + ".gx"], kb + ko[tag + ".gates"], kb + ko[tag + ".cs"]
                d.ws = sb + so[tag + ".ws"]
                d.hn_pos = pos_ptr[tag]
                d.x_absmax = kb + ko[tag + ".absmax"]
                d.B, d.T, d.I, d.H = B, Tn[tag], In[tag], H
            _lib.check(lib.mmb_bilstm_layer_fwd(descs, n, di, stream), "mmb_bilstm_layer_fwd")

        masks, held = {}, []
        if drop:
            # ONE generator call for the step's eleven masks (a flat vector of ones through F.dropout, cut by plan.mask_layout),
            # and one multi-tensor launch per stage to apply them -- 5 launches where eleven F.dropout calls and eleven products
            # took 22 (training mode at the metric configuration: 0.29 ms behind eval mode, most of it these kernels)
            flat = F.dropout(_ones_flat(plan.mask_total, dev), drop, True)
            masks = {name: flat[o:o + n].view(sh) for name, sh, o, n in plan.mask_layout}

        # ---- input encoders (models.py:97,102,113) + their output dropout (encoding.py:104)
        lstm_fwd(("et", "ea", "ei"), [xs["et"].data_ptr(), xs["ea"].data_ptr(), xs["ei"].data_ptr()])
        enc_out = {t: kb + ko[t + ".y"] for t in ("et", "ea", "ei")}
        att_in = {"aa": "ea", "ai": "ei"}
        att_d = {}
        if drop:
            ys = [view(ko[t + ".y"], (B, Tn[t], D)) for t in ("et", "ea", "ei")]
            yd = torch._foreach_mul(ys, [masks["out_et"], masks["out_ea"], masks["out_ei"]])
            held += yd
            enc_out = {"et": yd[0].data_ptr(), "ea": yd[1].data_ptr(), "ei": yd[2].data_ptr()}
            # dropped copies seen by the similarity only (attention.py:66-67)
            dd = torch._foreach_mul([yd[0], yd[1], yd[0], yd[2]], [masks["aa_t"], masks["aa_m"], masks["ai_t"], masks["ai_m"]])
            held += dd
            att_d = {"aa": (dd[0].data_ptr(), dd[1].data_ptr()), "ai": (dd[2].data_ptr(), dd[3].data_ptr())}
        # ---- the two attentions (models.py:131-132), one grouped call, shared text planes
        descs = (_lib.BidafDesc * 2)()
        for d, (tag, M) in zip(descs, plan.att):
            q = _P_ATT[tag]
            e = att_in[tag]
            d.text, d.mod = enc_out["et"], enc_out[e]
            d.text_mask = d.mod_mask = None
            d.text_len, d.mod_len = len_ptr["et"], len_ptr[e]
            if drop:
                d.text_d, d.mod_d = att_d[tag]
            else:
                d.text_d = d.mod_d = None
            d.w_t, d.w_m, d.w_tm, d.bias = pp[q], pp[q + 1], pp[q + 2], pp[q + 3]
            d.out, d.bsave = kb + ko[tag + ".out"], kb + ko[tag + ".bsave"]
            d.rterm, d.cterm, d.row_stat, d.col_stat = kb + ko[tag + ".rterm"], kb + ko[tag + ".cterm"], kb + ko[tag + ".rstat"], kb + ko[tag + ".cstat"]
            d.saved, d.saved_bytes = kb + ko[tag + ".saved"], plan.att_saved[tag]
            d.workspace, d.workspace_bytes = sb + so[tag + ".ws"], 256
            d.T, d.M = T, M
        _lib.check(lib.mmb_bidaf_group_fwd(descs, 2, B, D, di, stream), "mmb_bidaf_group_fwd")
        # ---- modelling encoders (models.py:134-135): layer 0, inter-layer dropout (encoding.py:81), layer 1, output dropout
        lstm_fwd(("a0", "i0"), [kb + ko["aa.out"], kb + ko["ai.out"]])
        l1_in = {"a1": kb + ko["a0.y"], "i1": kb + ko["i0.y"]}
        if drop:
            y0d = torch._foreach_mul([view(ko["a0.y"], (B, T, D)), view(ko["i0.y"], (B, T, D))], [masks["inter_a"], masks["inter_i"]])
            held += y0d
            l1_in = {"a1": y0d[0].data_ptr(), "i1": y0d[1].data_ptr()}
        lstm_fwd(("a1", "i1"), [l1_in["a1"], l1_in["i1"]])
        mod_out = [view(ko["a1.y"], (B, T, D)), view(ko["i1.y"], (B, T, D))]
        if drop:
            mod_out = torch._foreach_mul(mod_out, [masks["out_a"], masks["out_i"]])
        # ---- final hidden states (encoding.py:101-103) and the decoder's initial hidden state (models.py:143)
        hp = (ctypes.c_void_p * 4)(kb + ko["a0.hn"], kb + ko["a1.hn"], kb + ko["i0.hn"], kb + ko["i1.hn"])
        op = (ctypes.c_void_p * 2)(kb + ko["hid_a"], kb + ko["hid_i"])
        _lib.check(lib.mmb_hidden_states_fwd(hp, 2, 2, op, kb + ko["dec"], B, H, di, stream), "mmb_hidden_states_fwd")

        c = _Ctx()
        c.plan, c.meta, c.drop, c.masks, c.keep, c.xs, c.enc_out, c.l1_in = plan, meta, drop, masks, keep, xs, enc_out, l1_in
        c.att_d, c.held = att_d, held
        c.need_dx = [bool(ctx.needs_input_grad[1 + i]) for i in range(3)]
        ctx.c = c
        ctx.save_for_backward(*params)
        ctx.set_materialize_grads(False)
        return (mod_out[0], view(ko["hid_a"], (B, 4, H)), mod_out[1], view(ko["hid_i"], (B, 4, H)), view(ko["dec"], (B, H)))

    @staticmethod
    def backward(ctx, g_mod_a, g_hid_a, g_mod_i, g_hid_i, g_dec):
        lib = _lib.load()
        c = ctx.c
        params = ctx.saved_tensors
        plan, meta, drop, masks, keep = c.plan, c.meta, c.drop, c.masks, c.keep
        B, T, Ma, Mi, H = plan.dims
        D = 2 * H
        dev = keep.device
        di = dev.index
        main = torch.cuda.current_stream(dev)
        side = MF.side_stream(dev) if MF._USE_SIDE else main
        ms, ss = main.cuda_stream, side.cuda_stream
        kb = keep.data_ptr()
        ko = plan.keep.off
        bw = torch.empty(plan.bw.size, device=dev, dtype=torch.uint8)
        bb, bo = bw.data_ptr(), plan.bw.off
        mp = meta.data_ptr()
        len_ptr = {"et": mp, "ea": mp + 4 * B, "ei": mp + 8 * B, "a0": mp, "a1": mp, "i0": mp, "i1": mp}
        pos_ptr = {"et": mp + 12 * B, "ea": mp + 16 * B, "ei": mp + 20 * B, "a0": mp + 12 * B, "a1": mp + 12 * B, "i0": mp + 12 * B, "i1": mp + 12 * B}
        Tn = {"et": T, "ea": Ma, "ei": Mi, "a0": T, "a1": T, "i0": T, "i1": T}
        In = {"et": H, "ea": H, "ei": H, "a0": 8 * H, "a1": 2 * H, "i0": 8 * H, "i1": 2 * H}
        pp = [p.data_ptr() for p in params]
        # parameter gradients: fresh tensors shaped like the parameters (AccumulateGrad keeps what it is handed); the two bias
        # vectors of a direction have the same gradient but must not share storage: d_b and its twin
        g_wih = {t: torch.empty(2, 4 * H, In[t], device=dev, dtype=torch.float32) for t in _P_LSTM}
        g_whh = {t: torch.empty(2, 4 * H, H, device=dev, dtype=torch.float32) for t in _P_LSTM}
        g_b = torch.empty(2, 7, 2, 4 * H, device=dev, dtype=torch.float32)         # [copy][problem][direction][4H]
        g_att = torch.empty(2, 3 * D + 4, device=dev, dtype=torch.float32)
        lidx = {t: i for i, t in enumerate(("et", "ea", "ei", "a0", "i0", "a1", "i1"))}
        x_ptr = {"et": c.xs["et"].data_ptr(), "ea": c.xs["ea"].data_ptr(), "ei": c.xs["ei"].data_ptr(),
                 "a0": kb + ko["aa.out"], "i0": kb + ko["ai.out"], "a1": c.l1_in["a1"], "i1": c.l1_in["i1"]}

        def bview(off, shape):
            n = 1
            for s_ in shape:
                n *= s_
            return bw[off:off + 4 * n].view(torch.float32).view(shape)

        def cot(tag, g, shape, mask_key):
            """pointer of the cotangent of y[tag] the library reads: g itself, g * mask (training), or zeros"""
            if g is None:
                z = bview(bo[tag + ".d_y"], shape)
                z.zero_()
                return z.data_ptr(), z
            g = g.contiguous()
            if drop and mask_key is not None:
                o = bview(bo[tag + ".d_y"], shape)
                torch.mul(g, masks[mask_key], out=o)
                return o.data_ptr(), o
            return g.data_ptr(), g

        def lstm_descs(tags, d_y_ptrs, d_hn_ptrs, need_dx):
            n = len(tags)
            descs = (_lib.LstmBwdDesc * n)()
            for d, tag, dy, dhn, ndx in zip(descs, tags, d_y_ptrs, d_hn_ptrs, need_dx):
                q = _P_LSTM[tag]
                d.d_y, d.d_hn, d.x, d.y, d.lengths = dy, dhn, x_ptr[tag], kb + ko[tag + ".y"], len_ptr[tag]
                d.w_ih[0], d.w_ih[1], d.w_hh[0], d.w_hh[1] = pp[q], pp[q + 4], pp[q + 1], pp[q + 5]
                d.gates, d.cs = kb + ko[tag + ".gates"], kb + ko[tag + ".cs"]
                d.d_x = bb + bo[tag + ".d_x"] if ndx else None
                d.d_w_ih, d.d_w_hh = g_wih[tag].data_ptr(), g_whh[tag].data_ptr()
                d.d_b = g_b.data_ptr() + 4 * (lidx[tag] * 8 * H)
                d.d_a, d.d_w_cat, d.ws = bb + bo[tag + ".d_a"], bb + bo[tag + ".d_w_cat"], bb + bo[tag + ".ws"]
                d.hn_pos = pos_ptr[tag]
                d.x_absmax = kb + ko[tag + ".absmax"]
                d.B, d.T, d.I, d.H = B, Tn[tag], In[tag], H
            return descs, n

        def phase(dn, bits, stream_ptr, what):
            _lib.check(lib.mmb_bilstm_layer_bwd_phase(dn[0], dn[1], bits, di, stream_ptr), what)

        two = side is not main
        hold = []       # cotangent tensors that must outlive the enqueued kernels' host-side descriptors

        # ---- hidden states backward: per-layer d_h (B,2,H) in hn_pos order
        if g_hid_a is None and g_hid_i is None and g_dec is None:
            dh = {t: None for t in ("a0", "a1", "i0", "i1")}
        else:
            gh = [None if g is None else g.contiguous() for g in (g_hid_a, g_hid_i)]
            gd = None if g_dec is None else g_dec.contiguous()
            hold += [gh, gd]
            base = bb + bo["d_h"]
            step_b = B * 2 * H * 4
            dh = {"a0": base, "a1": base + step_b, "i0": base + 2 * step_b, "i1": base + 3 * step_b}
            gp = (ctypes.c_void_p * 2)(*[None if g is None else g.data_ptr() for g in gh])
            dp = (ctypes.c_void_p * 4)(dh["a0"], dh["a1"], dh["i0"], dh["i1"])
            _lib.check(lib.mmb_hidden_states_bwd(gp, None if gd is None else gd.data_ptr(), dp, 2, 2, B, H, di, ms), "mmb_hidden_states_bwd")

        # ---- modelling encoders, layer 1 (first recurrence of the pass: every layer's operand planes are prepared beside it)
        if drop and g_mod_a is not None and g_mod_i is not None:
            gy = torch._foreach_mul([g_mod_a.contiguous(), g_mod_i.contiguous()], [masks["out_a"], masks["out_i"]])
            pa, pi, ta, ti = gy[0].data_ptr(), gy[1].data_ptr(), gy[0], gy[1]
        else:
            pa, ta = cot("a1", g_mod_a, (B, T, D), "out_a")
            pi, ti = cot("i1", g_mod_i, (B, T, D), "out_i")
        hold += [ta, ti]
        L1 = lstm_descs(("a1", "i1"), (pa, pi), (dh["a1"], dh["i1"]), (True, True))
        L0 = lstm_descs(("a0", "i0"), (bb + bo["a1.d_x"], bb + bo["i1.d_x"]), (dh["a0"], dh["i0"]), (True, True))
        enc_tags = ("et", "ea", "ei")
        EN_prep = lstm_descs(enc_tags, (None, None, None), (None, None, None), c.need_dx)     # (PREPARE reads x, y, w_ih, x_absmax, ws, d_w_cat only)
        if two:
            before = torch.cuda.Event()
            before.record(main)
            phase(L1, 1 | HAVE_XC, ms, "bwd phase 1 (modelling layer 1)")
            side.wait_event(before)
            MF._side_head_start(di, side)
            for dn in (EN_prep, L0, L1):
                phase(dn, PREPARE | HAVE_WT, ss, "bwd prepare x planes")
            for dn, ndx in ((L0, (True, True)), (EN_prep, c.need_dx)):
                idx = [i for i, v in enumerate(ndx) if v]
                if idx:
                    sub = (_lib.LstmBwdDesc * len(idx))(*[dn[0][i] for i in idx])
                    phase((sub, len(idx)), PREPARE | HAVE_XC, ss, "bwd prepare w planes")
            prepared = torch.cuda.Event()
            prepared.record(side)
            f0 = HAVE_XC | HAVE_WT
        else:
            phase(L1, 1, ms, "bwd phase 1 (modelling layer 1)")
            phase(L1, 2, ms, "bwd phase 2 (modelling layer 1)")
            f0 = 0
        # inter-layer dropout backward, then layer 0
        if drop:
            torch._foreach_mul_([bview(bo["a1.d_x"], (B, T, D)), bview(bo["i1.d_x"], (B, T, D))], [masks["inter_a"], masks["inter_i"]])
        if two:
            main.wait_event(prepared)
            before = torch.cuda.Event()
            before.record(main)
            phase(L0, 1 | f0, ms, "bwd phase 1 (modelling layer 0)")
            side.wait_event(before)
            MF._side_head_start(di, side)
            phase(L1, 2 | HAVE_XC, ss, "bwd phase 2 (modelling layer 1)")
        else:
            phase(L0, 1, ms, "bwd phase 1 (modelling layer 0)")
            phase(L0, 2, ms, "bwd phase 2 (modelling layer 0)")
        # ---- attentions backward (full-chip kernels: nothing beside them)
        descs = (_lib.BidafDesc * 2)()
        att_in = {"aa": "ea", "ai": "ei"}
        for k, (d, (tag, M)) in enumerate(zip(descs, plan.att)):
            q = _P_ATT[tag]
            e = att_in[tag]
            d.text, d.mod = c.enc_out["et"], c.enc_out[e]
            d.text_mask = d.mod_mask = None
            d.text_len, d.mod_len = len_ptr["et"], len_ptr[e]
            if drop:
                d.text_d, d.mod_d = c.att_d[tag]
                d.d_text_d, d.d_mod_d = bb + bo[tag + ".d_text_d"], bb + bo[tag + ".d_mod_d"]
            else:
                d.text_d = d.mod_d = d.d_text_d = d.d_mod_d = None
            d.w_t, d.w_m, d.w_tm, d.bias = pp[q], pp[q + 1], pp[q + 2], None
            d.out, d.bsave = kb + ko[tag + ".out"], kb + ko[tag + ".bsave"]
            d.rterm, d.cterm, d.row_stat, d.col_stat = kb + ko[tag + ".rterm"], kb + ko[tag + ".cterm"], kb + ko[tag + ".rstat"], kb + ko[tag + ".cstat"]
            d.saved, d.saved_bytes = kb + ko[tag + ".saved"], plan.att_saved[tag]
            d.workspace, d.workspace_bytes = bb + bo[tag + ".ws"], plan.att_ws_b[tag]
            d.d_out = bb + bo[("a0" if tag == "aa" else "i0") + ".d_x"]
            d.d_text, d.d_mod = bb + bo[tag + ".d_text"], bb + bo[tag + ".d_mod"]
            gp_ = g_att.data_ptr() + 4 * k * (3 * D + 4)
            d.d_w_t, d.d_w_m, d.d_w_tm, d.d_bias = gp_, gp_ + 4 * D, gp_ + 8 * D, gp_ + 12 * D
            d.T, d.M = T, M
        _lib.check(lib.mmb_bidaf_group_bwd(descs, 2, B, D, di, ms), "mmb_bidaf_group_bwd")
        # cotangents of the input encoders' outputs: text gets both attentions' (+ the dropped copies' through their masks)
        d_text = bview(bo["aa.d_text"], (B, T, D))
        d_text.add_(bview(bo["ai.d_text"], (B, T, D)))
        d_aud, d_img = bview(bo["aa.d_mod"], (B, Ma, D)), bview(bo["ai.d_mod"], (B, Mi, D))
        if drop:
            # (d_text takes a term from each attention: two launches, a tensor must not appear twice in one multi-tensor update)
            for tag, M, dm in (("aa", Ma, d_aud), ("ai", Mi, d_img)):
                torch._foreach_addcmul_([d_text, dm], [bview(bo[tag + ".d_text_d"], (B, T, D)), bview(bo[tag + ".d_mod_d"], (B, M, D))],
                                        [masks[tag + "_t"], masks[tag + "_m"]])
            torch._foreach_mul_([d_text, d_aud, d_img], [masks["out_et"], masks["out_ea"], masks["out_ei"]])
        EN = lstm_descs(enc_tags, (d_text.data_ptr(), d_aud.data_ptr(), d_img.data_ptr()), (None, None, None), c.need_dx)
        if two:
            before = torch.cuda.Event()
            before.record(main)
            phase(EN, 1 | f0, ms, "bwd phase 1 (input encoders)")
            side.wait_event(before)
            MF._side_head_start(di, side)
            phase(L0, 2 | f0, ss, "bwd phase 2 (modelling layer 0)")
            phase(EN, 2 | f0, ms, "bwd phase 2 (input encoders)")
            main.wait_stream(side)
        else:
            phase(EN, 1, ms, "bwd phase 1 (input encoders)")
            phase(EN, 2, ms, "bwd phase 2 (input encoders)")
        g_b[1].copy_(g_b[0])
        # ---- hand the gradients back in param_list order
        grads = [None] * 64
        brow = g_b.view(28, 4 * H).unbind(0)          # [copy * 14 + problem * 2 + direction]
        for t, q in _P_LSTM.items():
            i = lidx[t]
            grads[q], grads[q + 4] = g_wih[t].unbind(0)
            grads[q + 1], grads[q + 5] = g_whh[t].unbind(0)
            grads[q + 2], grads[q + 6] = brow[2 * i], brow[2 * i + 1]
            grads[q + 3], grads[q + 7] = brow[14 + 2 * i], brow[14 + 2 * i + 1]
        for k, (tag, q) in enumerate(_P_ATT.items()):
            row = g_att[k]
            grads[q] = row[0:D].view(params[q].shape)
            grads[q + 1] = row[D:2 * D].view(params[q + 1].shape)
            grads[q + 2] = row[2 * D:3 * D].view(params[q + 2].shape)
            grads[q + 3] = row[3 * D:3 * D + 1].view(params[q + 3].shape)
        dxs = [bview(bo[t + ".d_x"], (B, Tn[t], H)) if nd else None for t, nd in zip(enc_tags, c.need_dx)]
        ctx.c = None
        return (None, *dxs, *grads)


class _Step:
    pass


def region_forward(R, x_text, x_aud, x_img, text_lengths, audio_lengths, image_lengths):
    """-> (mod_a, hid_a, mod_i, hid_i, dec_hidden (B,H)) through the single node; the caller has checked `eligible`."""
    drop = _drop_conf(R)
    if drop is False:
        return None
    B, T, H = x_text.shape
    st = _Step()
    st.plan = _plan(B, T, x_aud.shape[1], x_img.shape[1], H, drop is not None)
    st.meta = _meta(x_text.device, (text_lengths, audio_lengths, image_lengths))
    st.drop = drop
    lp = _last_params[0]
    ps = lp[1] if lp is not None and lp[0] == id(R) else param_list(R)      # (the list `eligible` has just checked)
    return _RegionFn.apply(st, x_text, x_aud, x_img, *ps)
