"""Batch-sharded data parallelism: one process per GPU, parameters replicated, ONE all-reduce of
a flat fp32 gradient buffer per step over RCCL/xGMI (backend "nccl" on ROCm) -- the MI355X-native
replacement of the reference's single-process nn.DataParallel (train.py:92).  The only exchange
step of the hot path; with the "gloo" backend the same code runs on CPU."""
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


class FlatGradAllReduce:
    """Averages the gradients of `params` across ranks with a single all-reduce of one flat buffer."""

    def __init__(self, params, group=None):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self.numel = sum(p.numel() for p in self.params)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        p0 = self.params[0]
        self.flat = torch.zeros(self.numel, dtype=torch.float32, device=p0.device)
        self.sizes = [p.numel() for p in self.params]

    def broadcast_parameters(self, src=0):
        """Make every replica start from rank `src`'s parameters (one flat broadcast)."""
        if self.world == 1:
            return
        with torch.no_grad():
            buf = torch.cat([p.detach().reshape(-1) for p in self.params])
            dist.broadcast(buf, src, group=self.group)
            for p, chunk in zip(self.params, buf.split(self.sizes)):
                p.copy_(chunk.view_as(p))

    def __call__(self):
        """grad <- mean over ranks (in place).  Parameters without a gradient contribute zeros."""
        if self.world == 1:
            return
        with torch.no_grad():
            torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in self.params],
                      out=self.flat)
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
            self.flat.mul_(1.0 / self.world)
            for p, chunk in zip(self.params, self.flat.split(self.sizes)):
                p.grad = chunk.view_as(p)
