"""The hot-path region of MMBiDAF as a stand-alone module (what bench.py / smoke() drive):
3 encoders -> 2 BiDAF attentions -> 2 two-layer modelling encoders, reference
models.py:47-78 (construction) and models.py:97,102,113,116-118,131-135,143 (forward), with the
same submodule names so that a reference state dict's hot-path entries load unchanged."""
import torch
import torch.nn as nn

from .attention import BiDAFAttention
from .encoding import RNNEncoder
from .model import MMBiDAF


class HotRegion(nn.Module):
    def __init__(self, hidden_size, drop_prob=0.):
        super().__init__()
        H = hidden_size
        self.text_enc = RNNEncoder(H, H, 1, drop_prob)
        self.audio_enc = RNNEncoder(H, H, 1, drop_prob)
        self.image_enc = RNNEncoder(H, H, 1, drop_prob)
        self.bidaf_att_audio = BiDAFAttention(2 * H, drop_prob=drop_prob)
        self.bidaf_att_image = BiDAFAttention(2 * H, drop_prob=drop_prob)
        self.mod_t_a = RNNEncoder(8 * H, H, 2, drop_prob)
        self.mod_t_i = RNNEncoder(8 * H, H, 2, drop_prob)

    get_mask = MMBiDAF.get_mask
    hot_path = MMBiDAF.hot_path

    def forward(self, x_text, x_aud, x_img, text_len, aud_len, img_len):
        """-> (mod_text_audio (B,T,2H), hidden_a (B,4,H), mod_text_image, hidden_i, decoder_hidden (B,1,H))"""
        mod_a, hid_a, mod_i, hid_i, _, dec_hidden = self.hot_path(x_text, x_aud, x_img, text_len, aud_len, img_len,
                                                                   with_decoder_hidden=True)
        return mod_a, hid_a, mod_i, hid_i, dec_hidden
