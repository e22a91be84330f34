"""Recipes shared by tools/gen_goldens.py (reference side) and the tests (our side) for the H=100 fixtures G7/G8.

Storing the 2.4 M hot-path parameters of an H=100 model (and as many gradients) would make a 20 MB fixture, so both
sides instead FILL the parameters by the same deterministic recipe, and big gradient matrices are compared through two
seeded random projections (G r and l^T G): an error in any element changes them.  The fixture stores a checksum of the
filled parameters so that a drift of torch's generator is noticed (then: re-run tools/gen_goldens.py)."""
import torch


def fill_parameters(named_params, seed, bound=0.1, skip=("image_keyframes_emb.",)):
    """p <- U(-bound, bound) from one torch.Generator(seed), in the order given (state-dict order on both sides).
    Returns (sum, sum of |.|) over everything filled."""
    g = torch.Generator().manual_seed(seed)
    tot, atot = 0.0, 0.0
    with torch.no_grad():
        for n, p in named_params:
            if any(n.startswith(s) for s in skip):
                continue
            v = (torch.rand(p.shape, generator=g, dtype=torch.float32) * 2 - 1) * bound
            p.copy_(v.to(p.device))
            tot += v.double().sum().item()
            atot += v.double().abs().sum().item()
    return tot, atot


def projections(name, grad, seed=4242):
    """(G r, l^T G) of a 2-D gradient with vectors seeded by the parameter name; 1-D gradients are kept whole."""
    if grad.dim() < 2:
        return {"full": grad.detach().cpu()}
    G = grad.detach().cpu().reshape(grad.shape[0], -1).double()
    g = torch.Generator().manual_seed(seed + sum(ord(c) for c in name))
    r = torch.randn(G.shape[1], generator=g, dtype=torch.float64)
    l = torch.randn(G.shape[0], generator=g, dtype=torch.float64)
    return {"right": (G @ r).float(), "left": (l @ G).float()}
