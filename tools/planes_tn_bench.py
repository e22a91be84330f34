#!/usr/bin/env python3
"""Weight-gradient shapes of the LSTM layers, C (M,N) = A^T (K,M)^T . B (N,K)^T: the k-major-A form of the operand-plane
GEMM (one row split of d_a serves the input and the weight gradient) next to the plain NT form (which needs a second,
transposing split of d_a).  Kernel-only times from the library's event hook; MMB_PLANES_TUNE applies.  GPU box only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mmbidaf_amd import functional as MF, _lib

dev = torch.device("cuda:0")
SHAPES = [("dW mod L0", 800, 1000, 12800), ("dW mod L1", 800, 400, 12800), ("dW text-enc", 800, 300, 12800), ("dW aud-enc", 800, 300, 8192)]
for name, M, N, K in SHAPES:
    at = torch.randn(K, M, device=dev)
    b = torch.randn(N, K, device=dev)
    a = at.t().contiguous()
    res = {}
    for form, fn in (("k-major A", lambda: MF.gemm_tn_planes(at, b)), ("NT", lambda: MF.gemm_nt_planes(a, b))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        _lib.profile_enable(["gemm", "split"])
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        _lib.profile_enable([])
        gms, gn, _ = _lib.profile_read("gemm")
        sms, sn, _ = _lib.profile_read("split")
        res[form] = (gms / gn * 1e3, sms / 10 * 1e3)
    fl = 2.0 * M * N * K
    print(f"{name:12s} {M}x{N}x{K}: " + " | ".join(f"{f}: gemm {g:7.1f} us ({3 * fl / (g * 1e-6) / 1e15:4.2f} PF) splits {s:6.1f} us" for f, (g, s) in res.items()), flush=True)
