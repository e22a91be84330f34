"""Batch-sharded data parallelism: one process per GPU, parameters replicated, the gradients reduced over
RCCL/xGMI (backend "nccl" on ROCm) through ONE flat fp32 buffer -- the MI355X-native replacement of the reference's
single-process nn.DataParallel (train.py:92).  The only exchange step of the hot path;
with the "gloo" backend the same code runs on CPU.

The buffer is cut into buckets that follow the order in which backward finishes the gradients (modelling encoders
first, input encoders last): with overlap=True each bucket's all-reduce is launched asynchronously from a
post-accumulate-grad hook as soon as its last gradient exists, so the exchange of the big modelling-encoder weights
(2.0 of the 2.4 M hot-path parameters) runs on RCCL's stream under the rest of the backward pass."""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:   # MMB_DIST_BACKEND=gloo: rehearsal of the N>1 path on a box with fewer GPUs than ranks
            backend = os.environ.get("MMB_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            local = local % torch.cuda.device_count()
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


def shard_range(global_batch, rank, world):
    """Samples [lo, hi) of the global batch owned by `rank` (even split)."""
    if global_batch % world:
        raise ValueError(f"global batch {global_batch} is not divisible by world size {world}")
    per = global_batch // world
    return rank * per, (rank + 1) * per


def region_buckets(region):
    """Parameter buckets of a HotRegion / MMBiDAF in the order backward completes them: second then first layer of the
    two modelling encoders (grouped launches, models.py:134-135), the two attentions, the three input encoders; any
    other trainable parameter (decoder, embeddings) goes first / last by position in the graph."""
    named = [(n, p) for n, p in region.named_parameters() if p.requires_grad]

    def pick(pred):
        return [p for n, p in named if pred(n)]
    is_mod = lambda n: n.startswith("mod_t_a.") or n.startswith("mod_t_i.")
    is_l1 = lambda n: "_l1" in n
    is_att = lambda n: n.startswith("bidaf_att_")
    is_enc = lambda n: n.split(".")[0] in ("text_enc", "audio_enc", "image_enc")
    is_dec = lambda n: n.startswith("multimodal_att_decoder.")
    order = [pick(is_dec), pick(lambda n: is_mod(n) and is_l1(n)), pick(lambda n: is_mod(n) and not is_l1(n)),
             pick(is_att), pick(is_enc),
             pick(lambda n: not (is_mod(n) or is_att(n) or is_enc(n) or is_dec(n)))]
    return [b for b in order if b]


class FlatGradAllReduce:
    """Reduces the gradients of `params` across ranks through one flat buffer.

    average=False (default): SUM.  The reference's loss is a SUM over the samples of the batch (models.py:168-176), so
    the gradient of the global batch is the sum of the shard gradients -- what one process on the whole batch (the
    reference's nn.DataParallel, train.py:92) computes, and what its clip threshold and step size see.
    average=True divides by the world size afterwards (mean-loss conventions).

    buckets: optional list of parameter lists (see region_buckets); default one bucket = one all-reduce.
    overlap=True: every bucket is packed and all-reduced asynchronously the moment backward has produced its last
    gradient (post-accumulate-grad hooks); __call__ then only launches what is still missing, waits and copies back.
    On the GPU the LSTM weight gradients are finished by DEFERRED side-stream work (functional.py, MMB_SIDE_STREAM=2), so a
    bucket launched from a hook must queue behind that work: defer_fn defaults to functional.defer_grad_work, which is
    correct in every side-stream mode (it runs the launch at once, behind the current stream, when nothing is deferred).

    The reduced values are copied back into the tensors autograd produced (p.grad is never rebound to a view of the
    flat buffer); a parameter without a gradient contributes zeros and keeps grad None."""

    def __init__(self, params, group=None, average=False, buckets=None, overlap=False, defer_fn=None):
        params = [p for p in params if p.requires_grad]
        if buckets is None:
            buckets = [params]
        else:
            buckets = [[p for p in b if p.requires_grad] for b in buckets]
            seen = {id(p) for b in buckets for p in b}
            rest = [p for p in params if id(p) not in seen]
            if rest:
                buckets = buckets + [rest]
        self.buckets = [b for b in buckets if b]
        self.params = [p for b in self.buckets for p in b]
        assert len({id(p) for p in self.params}) == len(self.params), "a parameter appears in two buckets"
        self.group = group
        self.average = average
        # defer_fn(device, fn): runs fn(stream) on the stream -- and at the moment -- the gradients of the bucket become
        # final (mmbidaf_amd.functional.defer_grad_work: the weight-gradient phase may itself be deferred to run beside the
        # next layer's recurrence; the bucket's packing + all-reduce are queued right behind it).  Default on CUDA with
        # overlap: exactly that function -- a hook that packed p.grad at once would read gradients the side stream has
        # not written yet.
        if defer_fn is None and overlap and params[0].is_cuda:
            from . import functional as _MF
            defer_fn = _MF.defer_grad_work
        self.defer_fn = defer_fn
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.numel = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(self.numel, dtype=torch.float32, device=self.params[0].device)
        self.slices, self.chunks = [], []
        o = 0
        for b in self.buckets:
            n = sum(p.numel() for p in b)
            sl = self.flat[o:o + n]
            self.slices.append(sl)
            self.chunks.append([c.view_as(p) for c, p in zip(sl.split([p.numel() for p in b]), b)])
            o += n
        self.work = [None] * len(self.buckets)
        self.pending = [0] * len(self.buckets)
        # optional timing (bench.py, N > 1): per call, the time the caller's stream spends inside __call__ (what the exchange EXPOSES
        # on the critical path: launches still missing, the wait for the collectives, the copy back) and, per bucket, from "packed"
        # to "reduced" (the all-reduce as the stream sees it) -- HIP events on the current stream, perf_counter on CPU tensors
        self.timing = False
        self._t_calls, self._t_buckets = [], []
        self.overlap = bool(overlap) and self.world > 1
        self._hooks = []
        if self.overlap:
            for bi, b in enumerate(self.buckets):
                for p in b:
                    self._hooks.append(p.register_post_accumulate_grad_hook(lambda p_, bi=bi: self._on_grad(bi)))
                    if self.defer_fn is not None:
                        p._mmb_deferral_aware = True     # functional._side_safe: these hooks queue behind deferred gradient work
            self._arm()

    def _arm(self):
        self.pending = [len(b) for b in self.buckets]

    def _on_grad(self, bi):
        self.pending[bi] -= 1
        if self.pending[bi] == 0 and self.work[bi] is None:
            self._launch(bi)

    def _launch(self, bi):
        if self.defer_fn is not None and self.flat.is_cuda:
            self.work[bi] = False        # claimed: the launch is queued behind the kernels that fill the bucket
            self.defer_fn(self.flat.device, lambda stream, bi=bi: self._launch_on(bi, stream))
        else:
            self._launch_here(bi)

    def _launch_on(self, bi, stream):
        with torch.cuda.stream(stream):
            self._launch_here(bi)

    def _stamp(self):
        if self.flat.is_cuda:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            return e
        import time
        return time.perf_counter()

    def _launch_here(self, bi):
        with torch.no_grad():
            grads = [p.grad for p in self.buckets[bi]]
            if all(g is not None for g in grads):
                torch.cat([g.reshape(-1) for g in grads], out=self.slices[bi])    # one packing kernel
            else:
                for g, c in zip(grads, self.chunks[bi]):
                    if g is None:
                        c.zero_()
                    else:
                        c.copy_(g)
            if self.timing:
                self._packed = getattr(self, "_packed", {})
                self._packed[bi] = self._stamp()
            self.work[bi] = dist.all_reduce(self.slices[bi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def broadcast_parameters(self, src=0):
        """Make every replica start from rank `src`'s parameters (one flat broadcast)."""
        if self.world == 1:
            return
        with torch.no_grad():
            buf = torch.cat([p.detach().reshape(-1) for p in self.params])
            dist.broadcast(buf, src, group=self.group)
            for p, chunk in zip(self.params, buf.split([p.numel() for p in self.params])):
                p.copy_(chunk.view_as(p))

    def __call__(self):
        """Call after backward: grad <- sum (or mean) over ranks, in place.
        Also the place where a time-out of a persistent recurrence launch of an EARLIER step surfaces: the
        gradients it would exchange are invalid, and inside a replayed graph no library call runs that could notice."""
        if self.flat.is_cuda:
            from . import _lib
            _lib.persist_check("FlatGradAllReduce")
        if self.world == 1:
            return
        t_in = self._stamp() if self.timing else None
        for bi in range(len(self.buckets)):
            if self.work[bi] is None:
                self._launch(bi)
        with torch.no_grad():
            for bi, b in enumerate(self.buckets):
                assert self.work[bi] is not False, "a deferred bucket launch did not run before the gradients were consumed"
                self.work[bi].wait()          # the current stream waits for the collective (no host sync with nccl)
                self.work[bi] = None
                if self.timing and bi in getattr(self, "_packed", {}):
                    self._t_buckets.append((self._packed.pop(bi), self._stamp()))
                if self.average:
                    self.slices[bi].mul_(1.0 / self.world)
                dst = [p.grad for p in b if p.grad is not None]
                src = [c for p, c in zip(b, self.chunks[bi]) if p.grad is not None]
                if dst:
                    torch._foreach_copy_(dst, src)
        if self.timing:
            self._t_calls.append((t_in, self._stamp()))
        if self.overlap:
            self._arm()

    def read_timing(self, reset=True):
        """-> (exposed_us_per_call, allreduce_us_per_call, calls): means over the calls made since timing was switched on / last
        read.  Synchronise the device first (the events must have completed)."""
        def span(a, b):
            return a.elapsed_time(b) * 1e3 if hasattr(a, "elapsed_time") else (b - a) * 1e6
        n = len(self._t_calls)
        exposed = sum(span(a, b) for a, b in self._t_calls) / n if n else None
        allred = sum(span(a, b) for a, b in self._t_buckets) / n if n else None
        if reset:
            self._t_calls, self._t_buckets = [], []
        return exposed, allred, n

    def remove_hooks(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
        for p in self.params:
            if hasattr(p, "_mmb_deferral_aware"):
                del p._mmb_deferral_aware
