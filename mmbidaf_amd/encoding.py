"""Drop-in for the reference's `layers.encoding` (same class and parameter names).

RNNEncoder runs on the HIP BiLSTM kernels (mmbidaf_amd.functional.bilstm_layer); Embedding /
HighwayEncoder / ImageEmbedding are the surrounding graph (stock PyTorch-ROCm).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import functional as MF


class HighwayEncoder(nn.Module):
    """num_layers highway layers  x <- g * relu(T x) + (1 - g) * x,  g = sigmoid(G x)
    (reference layers/encoding.py:33-59; parameter names `transforms.k`, `gates.k`)."""

    def __init__(self, num_layers, hidden_size):
        super().__init__()
        self.transforms = nn.ModuleList(nn.Linear(hidden_size, hidden_size) for _ in range(num_layers))
        self.gates = nn.ModuleList(nn.Linear(hidden_size, hidden_size) for _ in range(num_layers))

    def forward(self, x):
        for gate, transform in zip(self.gates, self.transforms):
            g = torch.sigmoid(gate(x))
            x = g * F.relu(transform(x)) + (1 - g) * x
        return x


class Embedding(nn.Module):
    """dropout -> bias-free projection to hidden_size -> 2-layer highway
    (reference layers/encoding.py:9-30; parameter names `proj`, `hwy`)."""

    def __init__(self, embedding_size, hidden_size, drop_prob):
        super().__init__()
        self.drop_prob = drop_prob
        self.proj = nn.Linear(embedding_size, hidden_size, bias=False)
        self.hwy = HighwayEncoder(2, hidden_size)

    def forward(self, x):
        x = F.dropout(x, self.drop_prob, self.training)
        H = self.proj.weight.shape[0]
        if x.is_cuda and H % 4 == 0:
            # one GEMM + one fused element-wise kernel per highway layer (mmbidaf_amd/functional.py)
            from . import functional as MF
            return MF.embedding_forward(x, self.proj.weight, self.hwy.gates, self.hwy.transforms)
        return self.hwy(self.proj(x))


class _DeviceCache:
    """Small LRU of host-derived index tensors already resident on the device (lengths, masks, sort order), keyed
    by the Python lengths they were built from.  A pageable H2D copy is stream-synchronous and stalls the host
    behind all queued kernels; on a miss the copy goes through pinned memory and is non-blocking."""

    def __init__(self, capacity=256):
        self.capacity = capacity
        self.items = {}

    def get(self, key, build):
        t = self.items.pop(key, None)
        if t is None:
            t = build()
            if len(self.items) >= self.capacity:
                self.items.pop(next(iter(self.items)))
        self.items[key] = t
        return t


_cache = _DeviceCache()


def to_device_cached(kind, lengths, device, build_host):
    """build_host() -> CPU tensor derived from `lengths`; returns it on `device` (cached per lengths tuple)."""
    key = (kind, str(device), tuple(lengths))

    def build():
        host = build_host()
        if device.type == "cuda":
            return host.pin_memory().to(device, non_blocking=True)
        return host.to(device)
    return _cache.get(key, build)


def sorted_order(lengths):
    """Descending-length order exactly as the reference computes it: float-cast lengths and
    torch.sort(descending=True) on the host (layers/encoding.py:85,91) -- tie order included (Q3)."""
    return torch.Tensor(lengths).sort(0, descending=True)[1]


def lstm_direction_params(rnn, layer):
    """[w_ih, w_hh, b_ih, b_hh] of the forward and of the reverse direction of `layer`."""
    names = ("weight_ih_l{}{}", "weight_hh_l{}{}", "bias_ih_l{}{}", "bias_hh_l{}{}")
    fwd = [getattr(rnn, n.format(layer, "")) for n in names]
    rev = [getattr(rnn, n.format(layer, "_reverse")) for n in names]
    return fwd, rev


def encode_group(encoders, xs, lengths_list, cat_hidden=True):
    """Run several independent RNNEncoders with the same depth and hidden size as ONE grouped
    launch per layer (models.py:97,102,113 and models.py:134-135 are independent of each other).
    Returns [(y, h_n_sorted)] exactly as each encoder's forward would; cat_hidden=False: h_n stays the list over layers of
    (B,2,H) tensors (for mmbidaf_amd.functional.hidden_states, which concatenates and sums them in one launch)."""
    n = len(encoders)
    L = encoders[0].rnn.num_layers
    assert all(e.rnn.num_layers == L and e.rnn.hidden_size == encoders[0].rnn.hidden_size for e in encoders)
    dev = xs[0].device
    lens_dev = [to_device_cached("len_i32", l, dev, lambda l=l: torch.tensor(list(l), dtype=torch.int32)) for l in lengths_list]
    for x, l in zip(xs, lengths_list):
        if len(l) != x.size(0) or min(l) < 1 or max(l) > x.size(1):
            raise ValueError("lengths must have one entry per sample with 1 <= len <= seq_len")
    # position of every sample in the reference's descending-length order (Q3): the kernels write h_n straight into
    # that order -- (B,2,H) per layer -- instead of a gather afterwards (and read its cotangent the same way)
    def inverse_order(lengths):
        order = sorted_order(lengths)
        inv = torch.empty_like(order)
        inv[order] = torch.arange(order.numel())
        return inv.to(torch.int32)
    pos_dev = [to_device_cached("hn_pos", l, dev, lambda l=l: inverse_order(l)) for l in lengths_list]
    inputs = list(xs)
    h_all = [[] for _ in range(n)]
    for k in range(L):
        problems = []
        for e, inp, ld, pd in zip(encoders, inputs, lens_dev, pos_dev):
            fwd, rev = lstm_direction_params(e.rnn, k)
            problems.append((inp, ld, fwd, rev, pd))
        outs = MF.bilstm_layer(problems)
        for i, (y, h_n) in enumerate(outs):
            h_all[i].append(h_n)
            e = encoders[i]
            if k < L - 1 and e.training and e.rnn.dropout > 0.0:   # nn.LSTM inter-layer dropout (Q7)
                y = F.dropout(y, e.rnn.dropout, True)
            inputs[i] = y
    results = []
    for e, y, hs, lengths in zip(encoders, inputs, h_all, lengths_list):
        y = F.dropout(y, e.drop_prob, e.training)                  # encoding.py:104, also for 1-layer encoders
        h_n = hs if not cat_hidden else hs[0] if len(hs) == 1 else torch.cat(hs, dim=1)   # (B,2L,H): [l0_fwd, l0_bwd, l1_fwd, l1_bwd] (Q4),
        results.append((y, h_n))                                   # rows already in length-sorted order (Q3)
    return results


class RNNEncoder(nn.Module):
    """Bidirectional multi-layer LSTM over padded variable-length sequences.

    Same constructor / forward signature and state-dict keys (`rnn.weight_ih_l0`, ...) as
    reference layers/encoding.py:62-108: `self.rnn` is a real nn.LSTM used as the parameter
    container (identical names, shapes and default initialisation), but the arithmetic runs in
    the HIP kernels.  forward(x (B,T,I), lengths: list[int]) -> (y (B,T,2H), h_n (B,2L,H)),
    with y in batch order and h_n in descending-length order, as the reference returns them.
    """

    def __init__(self, input_size, hidden_size, num_layers, drop_prob=0.):
        super().__init__()
        self.drop_prob = drop_prob
        self.rnn = nn.LSTM(input_size, hidden_size, num_layers, batch_first=True, bidirectional=True,
                           dropout=drop_prob if num_layers > 1 else 0.)

    def forward(self, x, lengths):
        return encode_group([self], [x], [lengths])[0]


class ImageEmbedding(nn.Module):
    """Frozen ResNet-101 feature extractor (reference layers/encoding.py:111-154).  Third-party
    CNN, not part of the hot path: the backbone is injectable (`backbone=`) and otherwise built
    lazily from torchvision, which must then be installed."""

    def __init__(self, backbone=None):
        super().__init__()
        if backbone is None:
            try:
                import torchvision
            except ImportError as e:
                raise RuntimeError("ImageEmbedding needs torchvision for its ResNet-101 backbone; "
                                   "pass ImageEmbedding(backbone=module) / MMBiDAF(..., image_backbone=module) "
                                   "to inject a feature extractor instead") from e
            backbone = torchvision.models.resnet101(pretrained=True)
        self.resnet = backbone
        self.fine_tune()

    def forward(self, images):
        return self.resnet(images)

    def fine_tune(self, fine_tune=False):
        for p in self.resnet.parameters():
            p.requires_grad = False
        for child in list(self.resnet.children())[5:]:
            for p in child.parameters():
                p.requires_grad = fine_tune
