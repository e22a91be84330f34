import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from mmbidaf_amd import functional as MF
d = torch.device("cuda:0")
MF.set_precision("bf16")
for M, N, K in ((128, 64, 32), (64, 320, 32), (64, 64, 128), (300, 400, 100)):
    g = torch.Generator().manual_seed(1)
    a, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g)
    ref = (a.bfloat16().double() @ b.bfloat16().double().t()).float()
    got = MF.gemm_nt_planes(a.to(d), b.to(d)).cpu()
    e = (got - ref).abs()
    print(M, N, K, "max err", e.max().item())
    for m0 in range(0, M, 16):
        row = ["%.0e" % e[m0:m0 + 16, n0:n0 + 16].max().item() for n0 in range(0, N, 16)]
        print("  rows %3d:" % m0, " ".join(row))
