#!/usr/bin/env python3
"""PRICING PROBE (GPU box): do two half batches, issued as two independent chains on two streams inside one captured step,
finish sooner than the one full-batch chain?  The recurrences are latency-bound (one workgroup per (encoder, direction, sample):
192 / 128 / 128 workgroups on 256 CUs, their duration does not depend on B), everything between them is throughput-bound; two
chains half a phase apart could run one chain's GEMMs / attention beside the other's recurrences.

    MMB_SIDE_GATE=0 python tools/half_batch_probe.py [--steps 60]

Timing only: the two chains use two module copies (their own gradients) and each its own side stream; the per-device dispatch gate
is off (it is one word per device).  Prints ms per step of: one B=32 chain; one B=16 chain; two B=16 chains started `stagger` us apart.

Round 5 result (profiles/r05_half_batch_probe.txt): one B=32 chain 2.250 ms, one B=16 chain 1.910 ms (the recurrences do not get
shorter); the two-chain capture ends in a segmentation fault inside hipStreamEndCapture on this ROCm build (a second chain forked
from the capturing stream with its own forked side stream), so the concurrent figure was not measured.  Upper bound of the idea:
both chains done at 1.91 ms + stagger with perfect overlap, i.e. <= 7-9 %; not pursued.
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MMB_SIDE_GATE", "0")
import torch  # noqa: E402

from mmbidaf_amd import _lib, synth  # noqa: E402
from mmbidaf_amd import functional as MF  # noqa: E402
from mmbidaf_amd.hot_region import HotRegion  # noqa: E402


def build(B, dev, seed=224):
    torch.manual_seed(seed)
    region = HotRegion(100).to(dev)
    region.eval()
    batch = synth.make_batch("cfg2", device=dev, batch=B)
    xs = [batch[k].requires_grad_(True) for k in ("x_text", "x_aud", "x_img")]
    return region, batch, xs


def one_step(region, batch, xs):
    outs = region(*xs, batch["text_len"], batch["aud_len"], batch["img_len"])
    synth.region_loss(outs, batch).backward()


def clear(chains):
    for region, _, xs in chains:
        for p in region.parameters():
            p.grad = None
        for x in xs:
            x.grad = None


def capture(chains, stagger_us, dev):
    """chains: list of (region, batch, xs); chain k runs on its own stream (and side stream), started k * stagger_us late."""
    di = dev.index or 0
    streams = [torch.cuda.Stream() for _ in chains]
    sides = [torch.cuda.Stream() for _ in chains]
    lib = _lib.load()

    def issue():
        cur = torch.cuda.current_stream()
        cur0 = torch.cuda.Stream()          # forks from the START of the step (an event recorded before chain 0 is issued)
        cur0.wait_stream(cur)
        for k, (c, s) in enumerate(zip(chains, streams)):
            MF._side_streams[di] = sides[k]
            if k == 0:          # chain 0 on the capturing stream itself
                one_step(*c)
                continue
            s.wait_stream(cur0)
            with torch.cuda.stream(s):
                if stagger_us:
                    _lib.check(lib.mmb_stream_delay(di, s.cuda_stream, int(k * stagger_us)), "delay")
                one_step(*c)
        for s in streams[1:]:
            cur.wait_stream(s)
        cur.wait_stream(cur0)

    torch.cuda.synchronize()
    warm = torch.cuda.Stream()
    warm.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(warm):
        for _ in range(3):
            clear(chains)
            issue()
    torch.cuda.current_stream().wait_stream(warm)
    torch.cuda.synchronize()
    clear(chains)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        issue()
    return g


def timeit(g, steps):
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--staggers", default="0,60,120,200,300")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    full = build(32, dev)
    print(f"one chain  B=32            {timeit(capture([full], 0, dev), a.steps):7.3f} ms/step", flush=True)
    h1, h2 = build(16, dev), build(16, dev, seed=225)
    print(f"one chain  B=16            {timeit(capture([h1], 0, dev), a.steps):7.3f} ms/step", flush=True)
    for st in [int(s) for s in a.staggers.split(",")]:
        print(f"two chains B=16 +{st:4d} us   {timeit(capture([h1, h2], st, dev), a.steps):7.3f} ms/step", flush=True)
    q = [build(8, dev, seed=230 + k) for k in range(4)]
    for st in (0, 60, 120):
        print(f"four chains B=8 +{st:4d} us   {timeit(capture(q, st, dev), a.steps):7.3f} ms/step", flush=True)


if __name__ == "__main__":
    main()
