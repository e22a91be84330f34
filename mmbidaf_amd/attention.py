"""Drop-in for the reference's `layers.attention` (same class / function / parameter names).

BiDAFAttention runs on the fused HIP kernels (mmbidaf_amd.functional.bidaf_attention);
masked_softmax and MultimodalAttentionDecoder are the surrounding graph (stock PyTorch-ROCm,
SURVEY.md section 8(f)) and keep the reference's parameter names so its checkpoints load.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import functional as MF


class BiDAFAttention(nn.Module):
    """Bidirectional attention flow between the text and one other modality.

    Same constructor, parameters (`text_weight (D,1)`, `modality_weight (D,1)`,
    `text_modality_weight (1,1,D)`, `bias (1,)`), initialisation and forward signature as
    reference layers/attention.py:9-54.  forward -> (B, T, 4*D) = [text, a, text*a, text*b].
    """

    def __init__(self, hidden_size, drop_prob=0.1):
        super().__init__()
        self.drop_prob = drop_prob
        self.text_weight = nn.Parameter(torch.zeros(hidden_size, 1))
        self.modality_weight = nn.Parameter(torch.zeros(hidden_size, 1))
        self.text_modality_weight = nn.Parameter(torch.zeros(1, 1, hidden_size))
        for w in (self.text_weight, self.modality_weight, self.text_modality_weight):
            nn.init.xavier_uniform_(w)
        self.bias = nn.Parameter(torch.zeros(1))

    def _dropped(self, text, modality):
        # only the similarity sees the dropped copies (reference attention.py:66-67, quirk Q6);
        # the masks come from torch's RNG on the host side of the boundary
        if self.training and self.drop_prob > 0.0:
            return F.dropout(text, self.drop_prob, True), F.dropout(modality, self.drop_prob, True)
        return None, None

    def forward(self, text, modality, text_mask, modality_mask):
        text_d, mod_d = self._dropped(text, modality)
        return MF.bidaf_attention(text, modality, text_mask, modality_mask, self.text_weight,
                                  self.modality_weight, self.text_modality_weight, self.bias,
                                  text_d=text_d, mod_d=mod_d)

    @staticmethod
    def forward_group(modules, texts, modalities, text_masks, modality_masks):
        """Several independent BiDAFAttention modules (the model's text<->audio / text<->image pair, reference
        models.py:131-132) in ONE grouped library call: one launch per stage, text planes shared when the text tensor is."""
        import torch.nn.modules.module as _M
        hooked = any(m._forward_hooks or m._forward_pre_hooks or m._backward_hooks or m._backward_pre_hooks for m in modules) or \
            _M._global_forward_hooks or _M._global_forward_pre_hooks or _M._global_backward_hooks
        if hooked:       # module hooks only fire through Module.__call__: keep nn.Module semantics, one call per module
            return [m(t, x, tm, mm) for m, t, x, tm, mm in zip(modules, texts, modalities, text_masks, modality_masks)]
        problems = []
        for m, text, mod, tm, mm in zip(modules, texts, modalities, text_masks, modality_masks):
            text_d, mod_d = m._dropped(text, mod)
            problems.append((text, mod, tm, mm, m.text_weight, m.modality_weight, m.text_modality_weight, m.bias, text_d, mod_d))
        return MF.bidaf_attention_group(problems)

    def get_similarity_matrix(self, text, modality):
        """(B,T,M) trilinear similarity (reference attention.py:56-75).  Kept for API parity; the
        fused forward never materialises it, so this is plain tensor algebra on whatever device
        the inputs are on."""
        text_d, mod_d = self._dropped(text, modality)
        t = text if text_d is None else text_d
        m = modality if mod_d is None else mod_d
        row = t @ self.text_weight                                   # (B,T,1)
        col = (m @ self.modality_weight).transpose(1, 2)             # (B,1,M)
        cross = torch.bmm(t * self.text_modality_weight, m.transpose(1, 2))
        return row + col + cross + self.bias


def masked_softmax(logits, mask, dim=-1, log_softmax=False):
    """softmax over `dim` of mask*logits + (1-mask)*(-1e30) (reference attention.py:78-98);
    a fully masked slice therefore comes out uniform, not NaN."""
    keep = mask.type(torch.float32)
    blended = keep * logits + (1.0 - keep) * -1e30
    return (F.log_softmax if log_softmax else F.softmax)(blended, dim)


class MultimodalAttentionDecoder(nn.Module):
    """Pointer-style decoder step with coverage over the two modality-aware encodings
    (reference layers/attention.py:100-186).  Surrounding graph: stock PyTorch ops."""

    def __init__(self, text_embedding_size, hidden_size, output_size, num_layers=1, dropout=0.1):
        super().__init__()
        self.text_embedding_size = text_embedding_size
        self.hidden_size = hidden_size
        self.output_size = output_size
        self.num_layers = num_layers
        self.dropout = dropout
        H2 = 2 * hidden_size
        # text<->audio additive attention
        self.W1 = nn.Linear(H2, H2)
        self.W2 = nn.Linear(hidden_size, H2)
        self.Wc1 = nn.Linear(1, H2)
        self.v1 = nn.Linear(H2, 1)
        self.tanh = nn.Tanh()
        # text<->image additive attention
        self.W3 = nn.Linear(H2, H2)
        self.W4 = nn.Linear(hidden_size, H2)
        self.Wc2 = nn.Linear(1, H2)
        self.v2 = nn.Linear(H2, 1)
        # gate between the two contexts
        self.W_beta_1 = nn.Linear(H2, H2)
        self.W_beta_2 = nn.Linear(hidden_size, H2)
        self.W_beta_3 = nn.Linear(H2, H2)
        self.W_beta_4 = nn.Linear(hidden_size, H2)
        self.v_beta_1 = nn.Linear(H2, 1)
        self.v_beta_2 = nn.Linear(H2, 1)
        # output
        self.lstm = nn.LSTM(text_embedding_size + H2, hidden_size, num_layers, batch_first=True)
        self.out = nn.Linear(hidden_size, output_size)
        self.softmax = nn.Softmax()

    @staticmethod
    def _context(memory, proj_mem, proj_hid, proj_cov, score, hidden, coverage):
        weights = F.softmax(score(torch.tanh(proj_mem(memory) + proj_hid(hidden) + proj_cov(coverage))), dim=1)
        return weights, (weights * memory).sum(dim=1)

    def forward(self, sent_embed, decoder_hidden, decoder_cell_state, text_audio_enc_out, text_img_enc_out,
                coverage_vec, mask):
        w_a, ctx_a = self._context(text_audio_enc_out, self.W1, self.W2, self.Wc1, self.v1, decoder_hidden, coverage_vec)
        w_i, ctx_i = self._context(text_img_enc_out, self.W3, self.W4, self.Wc2, self.v2, decoder_hidden, coverage_vec)
        gate_a = self.v_beta_1(torch.tanh(self.W_beta_1(ctx_a.unsqueeze(1)) + self.W_beta_2(decoder_hidden)))
        gate_i = self.v_beta_2(torch.tanh(self.W_beta_3(ctx_i.unsqueeze(1)) + self.W_beta_4(decoder_hidden)))
        beta = F.softmax(torch.cat((gate_a, gate_i), dim=1), dim=1)                 # (B,2,1)
        fused = (torch.stack((ctx_a, ctx_i), dim=1) * beta).sum(dim=1)              # (B,2H)
        att_cov_dist = torch.bmm(torch.cat((w_a, w_i), dim=2), beta)                # (B,T,1)
        coverage_vec = coverage_vec + att_cov_dist
        step_in = torch.cat((fused.unsqueeze(1), sent_embed), dim=2)
        dec_out, (decoder_hidden, decoder_cell_state) = self.lstm(
            step_in, (decoder_hidden.transpose(0, 1).contiguous(), decoder_cell_state))
        dec_out = dec_out.reshape(-1, dec_out.size(-1))
        final_out = masked_softmax(self.out(dec_out), mask)
        return final_out, decoder_hidden.transpose(0, 1), decoder_cell_state, att_cov_dist, coverage_vec
