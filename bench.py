#!/usr/bin/env python3
"""Benchmark of the MMBiDAF hot-path region on MI355X (BASELINE.json metric):

    samples/sec forward+backward of  3 BiLSTM encoders -> 2 BiDAF attentions -> 2 two-layer
    modelling encoders  on synthetic (B, T_text=400, T_aud=256, T_img=64, H=100) batches.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A step = one forward+backward pass of the region over one batch already resident in HBM
(plus, for N > 1, the flat-gradient all-reduce over RCCL).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

from mmbidaf_amd import _lib, ddp, synth
from mmbidaf_amd.hot_region import HotRegion

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP32_MFMA_PEAK_TF = 157.3    # exact-f32 MFMA / vector peak
ATT_FWD_KERNELS = ["att_rank1", "att_col", "att_combine", "att_row"]
ATT_BWD_KERNELS = ["att_bwd_pre", "att_bwd_j1", "att_bwd_j2", "att_bwd_jfin", "att_bwd_i"]
ALL_KERNELS = ATT_FWD_KERNELS + ATT_BWD_KERNELS + ["gemm", "lstm_rec_fwd", "lstm_rec_bwd"]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="cfg2", choices=sorted(synth.CONFIGS))
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch override")
    ap.add_argument("--ragged", action="store_true", help="lengths ~U{n/2..n} instead of full")
    ap.add_argument("--fresh-lengths", action="store_true",
                    help="new ragged lengths every step (the host-derived masks / sort orders miss the device cache, as in real training)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-all", action="store_true", help="time every kernel class (perturbs the step time a little)")
    return ap.parse_args()


def cpu_baseline(region, cfg, ragged):
    """The oracle's CPU path (torch's own packed nn.LSTM + bmm/softmax attention, i.e. what the
    reference's modules execute) timed on the host cores on a bounded sample of the workload."""
    from oracle import mmbidaf_oracle as O
    B_s = 8
    batch = synth.make_batch(cfg, rank=0, ragged=ragged, device="cpu", batch=B_s)
    H = batch["H"]
    threads = min(16, len(os.sched_getaffinity(0)))   # the GPU box gives one GPU a 16-core share
    torch.set_num_threads(threads)
    ref = O.HotRegionCPU({k: v.detach().cpu() for k, v in region.state_dict().items()}, H)
    xs = [batch[k].requires_grad_(True) for k in ("x_text", "x_aud", "x_img")]

    def step():
        ref.zero_grad(set_to_none=True)
        for x in xs:
            x.grad = None
        outs = ref(*xs, batch["text_len"], batch["aud_len"], batch["img_len"])
        synth.region_loss(outs, batch).backward()

    step()
    n, t0 = 0, time.perf_counter()
    while n < 2 or (time.perf_counter() - t0 < 10.0 and n < 10):
        step()
        n += 1
    dt = time.perf_counter() - t0
    return {"value": round(B_s * n / dt, 3), "unit": "samples/s", "cores": threads, "kind": "port",
            "sample": f"{n} fwd+bwd steps of the same workload at batch {B_s} (of {synth.CONFIGS[cfg][0]}) after 1 warm-up, "
                      f"oracle HotRegionCPU = torch {torch.__version__} CPU packed nn.LSTM + bmm/softmax attention, "
                      f"{threads} threads; the reference's own Python does not travel to this box"}


def main():
    a = parse()
    rank, world, local = ddp.init_from_env()
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}")
    assert torch.cuda.is_available(), "bench.py needs an MI355X; there is no CPU fallback for the hot path"
    local = local % torch.cuda.device_count()   # (rehearsals with more ranks than GPUs share a device)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    _lib.load()

    B, T, Ma, Mi, H = synth.CONFIGS[a.config]
    if a.batch:
        B = a.batch
    D = 2 * H
    torch.manual_seed(224)  # the reference's seed (args.py:45): identical replicas on every rank
    region = HotRegion(H).to(dev)
    params = list(region.parameters())
    sync = ddp.FlatGradAllReduce(params)
    sync.broadcast_parameters()
    batch = synth.make_batch(a.config, rank=rank, ragged=a.ragged, device=dev, batch=B)
    xs = [batch[k].requires_grad_(True) for k in ("x_text", "x_aud", "x_img")]  # they come from trainable embeddings

    import random
    rng = random.Random(99 + rank)

    def step():
        for p in params:
            p.grad = None
        for x in xs:
            x.grad = None
        if a.fresh_lengths:
            for key, n in (("text_len", T), ("aud_len", Ma), ("img_len", Mi)):
                batch[key] = [n] + [rng.randint(max(1, n // 2), n) for _ in range(B - 1)]
        outs = region(*xs, batch["text_len"], batch["aud_len"], batch["img_len"])
        synth.region_loss(outs, batch).backward()
        sync()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    timed = ALL_KERNELS if a.profile_all else ATT_FWD_KERNELS
    fence()
    _lib.profile_enable(timed)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    _lib.profile_enable([])
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    prof = {k: _lib.profile_read(k) for k in timed}

    if rank == 0:
        ms_row, n_row, sym = prof["att_row"]
        # dominant attention kernel = the row pass; its launches alternate text<->audio / text<->image
        bytes_row = sum(4 * B * (5 * T * D + 2 * M * D) for M in (Ma, Mi)) / 2.0   # per launch, averaged over the two
        flops_row = sum(2 * B * T * M * (208 + 2 * 208) for M in (Ma, Mi)) / 2.0     # S + 2 PV products at the padded D
        # (PMC `traffic`: profiles/pmc_traffic.json = FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE per launch, from separate
        #  rocprofv3 --pmc passes of this same command: tools/profile_summary.py)
        avg_s = ms_row / max(n_row, 1) * 1e-3
        achieved = bytes_row / avg_s / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(a.config, {}).get(sym)
            except Exception:
                traffic = None
        fwd_ms = sum(prof[k][0] for k in ATT_FWD_KERNELS) / max(a.steps, 1)
        fwd_bytes = sum(synth.attention_algorithmic_bytes(B, T, M, D) for M in (Ma, Mi))
        out = {
            "metric": "samples/sec fwd+bwd, synthetic T_text=400 H=100, at 1/2/4/8 MI355X",
            "value": round(world * B * a.steps / dt, 2), "unit": "samples/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "arithmetic": "fp32 throughout: attention and recurrences on exact-f32 MFMA / VALU; LSTM projection and gradient GEMMs "
                          "on fp16 MFMA from an error-compensated split of the fp32 operands (two fp16 terms of the power-of-two-scaled "
                          "value, 3 cross products, fp32 accumulate: max error vs float64 3-8e-7 of the output scale, the same as an fp32 GEMM)",
            "config": {"workload": f"{a.config}: hot-path region (3 BiLSTM enc -> 2 BiDAF att -> 2 two-layer BiLSTM) "
                                   f"B={B}/GPU T_text={T} T_aud={Ma} T_img={Mi} H={H}, "
                                   f"{'ragged U{n/2..n}' if a.ragged else 'full'} lengths, fwd+bwd"
                                   f"{' + flat-grad all-reduce' if world > 1 else ''}",
                       "global_batch": world * B, "parallelism": f"dp{world}"},
            "roofline": {"bound": "hbm", "kernel": sym, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "avg_launch_us": round(avg_s * 1e6, 2), "launches": n_row,
                         "algorithmic_bytes_per_launch": int(bytes_row),
                         "mfma_f32": {"achieved_tflops": round(flops_row / avg_s / 1e12, 2), "peak_tflops": FP32_MFMA_PEAK_TF,
                                      "frac": round(flops_row / avg_s / 1e12 / FP32_MFMA_PEAK_TF, 4)},
                         "fused_fwd_both_attentions": {"ms_per_step": round(fwd_ms, 4), "algorithmic_bytes": int(fwd_bytes),
                                                       "achieved_GBs": round(fwd_bytes / (fwd_ms * 1e-3) / 1e9, 1) if fwd_ms else None}},
        }
        if a.profile_all:
            out["kernel_ms_per_step"] = {k: round(v[0] / a.steps, 4) for k, v in prof.items()}
            out["kernel_launches_per_step"] = {k: v[1] / a.steps for k, v in prof.items()}
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(region, a.config, a.ragged)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
