#!/usr/bin/env python3
"""Microbenchmark of mmb_gemm_f32 on the GEMM shapes of the cfg2 hot path, next to torch.mm (rocBLAS /
hipBLASLt fp32) as a yardstick.  GPU box only."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from mmbidaf_amd import functional as MF
from mmbidaf_amd import _lib

dev = torch.device("cuda:0")
SHAPES = [  # (name, M, N, K, ta, tb)
    ("gx enc   NT", 12800, 400, 100, 0, 1),
    ("gx modL0 NT", 12800, 400, 800, 0, 1),
    ("gx modL1 NT", 12800, 400, 200, 0, 1),
    ("dx enc   NN", 12800, 100, 400, 0, 0),
    ("dx modL0 NN", 12800, 800, 400, 0, 0),
    ("dx modL1 NN", 12800, 200, 400, 0, 0),
    ("dWih enc TN", 800, 100, 12800, 1, 0),
    ("dWih L0  TN", 800, 800, 12800, 1, 0),
    ("dWih L1  TN", 800, 200, 12800, 1, 0),
    ("dWhh     TN", 400, 100, 12800, 1, 0),
]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


for name, M, N, K, ta, tb in SHAPES:
    a = torch.randn((K, M) if ta else (M, K), device=dev)
    b = torch.randn((N, K) if tb else (K, N), device=dev)
    aa = a.t() if ta else a
    bb = b.t() if tb else b
    ref = aa @ bb
    got = MF.gemm(a, b, ta=bool(ta), tb=bool(tb))
    err = (got - ref).abs().max().item() / max(1.0, ref.abs().max().item())
    t_mine = timeit(lambda: MF.gemm(a, b, ta=bool(ta), tb=bool(tb)))
    t_ref = timeit(lambda: torch.mm(aa, bb))
    fl = 2.0 * M * N * K
    print(f"{name}  {M:6d}x{N:4d}x{K:6d}  mine {t_mine*1e6:8.1f} us {fl/t_mine/1e12:6.1f} TF | torch.mm {t_ref*1e6:8.1f} us "
          f"{fl/t_ref/1e12:6.1f} TF | relerr {err:.1e}")

print("== operand-plane path (NT form of the same products; kernel-only time from the library's event hook)")
for name, M, N, K, ta, tb in SHAPES:
    if ta:   # weight-gradient shapes run as NT on transposed planes: (M, K) x (N, K)
        pass
    a = torch.randn(M, K, device=dev)
    b = torch.randn(N, K, device=dev)
    ref = a @ b.t()
    got = MF.gemm_nt_planes(a, b)
    err = (got - ref).abs().max().item() / max(1.0, ref.abs().max().item())
    for _ in range(3):
        MF.gemm_nt_planes(a, b)
    torch.cuda.synchronize()
    _lib.profile_enable(["gemm", "split"])
    for _ in range(10):
        MF.gemm_nt_planes(a, b)
    torch.cuda.synchronize()
    _lib.profile_enable([])
    gms, gn, _ = _lib.profile_read("gemm")
    sms, sn, _ = _lib.profile_read("split")
    fl = 2.0 * M * N * K
    t = gms / gn * 1e-3
    print(f"{name}  {M:6d}x{N:4d}x{K:6d}  planes gemm {t*1e6:8.1f} us {fl/t/1e12:6.1f} TF (+ splits {sms/10*1e3:6.1f} us) | relerr {err:.1e}")
