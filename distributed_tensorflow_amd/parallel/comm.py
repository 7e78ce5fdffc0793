"""Collective-communication layer: one process per GPU, ``torch.distributed`` with the
``nccl`` backend (RCCL over xGMI on ROCm) for GPU tensors and ``gloo`` for the CPU test path.

The reference moves every tensor through TF's gRPC ``RecvTensor`` (one RPC per variable per
direction, /root/reference/example.py:128-129,189); here the data plane is a handful of
large collectives / point-to-point transfers on flat buffers.
"""
from __future__ import annotations

import datetime
import os
from typing import List, Optional

import torch
import torch.distributed as dist


def env_rank() -> int:
    return int(os.environ.get("RANK", "0"))


def env_world_size() -> int:
    return int(os.environ.get("WORLD_SIZE", "1"))


def env_local_rank() -> int:
    return int(os.environ.get("LOCAL_RANK", os.environ.get("RANK", "0")))


_WARNED_SHARED = False


def pick_backend(device=None) -> str:
    """``nccl`` (RCCL) for GPU tensors, ``gloo`` for CPU.  ``DTF_COMM_BACKEND=gloo`` forces gloo for GPU tensors too:
    RCCL refuses two ranks on one device, so several ranks time-sharing a single GPU (the one-GPU test box) talk
    through gloo while every kernel still runs on the GPU."""
    forced = os.environ.get("DTF_COMM_BACKEND")
    if forced:
        return forced
    from ..cluster import current_server
    srv = current_server()
    if srv is not None and device is not None and torch.device(device).type == "cuda" and srv.gpus_oversubscribed():
        global _WARNED_SHARED
        if not _WARNED_SHARED:
            _WARNED_SHARED = True
            print("WARNING: %d worker tasks share %d GPU(s); RCCL needs one GPU per rank, so the collectives of this "
                  "run go over gloo (staged through host memory)" % (srv.num_workers, torch.cuda.device_count()), flush=True)
        return "gloo"
    if device is not None:
        return "nccl" if torch.device(device).type == "cuda" else "gloo"
    return "nccl" if torch.cuda.is_available() else "gloo"


def init_process_group(backend: Optional[str] = None, rank: Optional[int] = None, world_size: Optional[int] = None,
                       master_addr: Optional[str] = None, master_port: Optional[int] = None,
                       timeout_s: float = 600.0, device: Optional[torch.device] = None):
    """Initialise the default process group from arguments or torchrun-style environment variables."""
    if dist.is_initialized():
        return dist.group.WORLD
    rank = env_rank() if rank is None else rank
    world_size = env_world_size() if world_size is None else world_size
    if backend is None or os.environ.get("DTF_COMM_BACKEND"):
        backend = pick_backend(device)
    addr = master_addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
    port = master_port or int(os.environ.get("MASTER_PORT", "29500"))
    if os.environ.get("DTF_KEEP_IPC_MODE") != "1":
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC is what the host driver supports
    kwargs = {}
    if backend == "nccl":
        if device is None:
            device = torch.device("cuda", env_local_rank() % max(torch.cuda.device_count(), 1))
        torch.cuda.set_device(device)
        if os.environ.get("DTF_PG_LAZY_INIT") != "1":
            kwargs["device_id"] = device     # eager communicator creation; lazy = at the first collective
    dist.init_process_group(backend=backend, init_method="tcp://%s:%d" % (addr, port), rank=rank,
                            world_size=world_size, timeout=datetime.timedelta(seconds=timeout_s), **kwargs)
    return dist.group.WORLD


class Communicator:
    """Thin wrapper that papers over backend capability differences (gloo has no reduce-scatter)."""

    def __init__(self, group=None):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised; call init_process_group() first")
        self.group = group if group is not None else dist.group.WORLD
        self.rank = dist.get_rank(self.group)
        self.world_size = dist.get_world_size(self.group)
        self.backend = dist.get_backend(self.group)

    def _staged(self, t: torch.Tensor) -> bool:
        """GPU tensors over gloo (ranks time-sharing one GPU) are staged through host memory here, synchronously:
        gloo's own device path runs copies on side streams from its worker thread, which has been seen to stall
        when several freshly started processes share one GPU."""
        return self.backend == "gloo" and t.is_cuda

    # every method returns a Work handle when async_op=True
    def reduce_scatter_sum(self, out_shard: torch.Tensor, full: torch.Tensor, async_op: bool = False):
        """``out_shard = sum over ranks of full[rank*n:(rank+1)*n]``; ``out_shard`` may alias that slice."""
        if self.world_size == 1:
            if out_shard.data_ptr() != full.data_ptr():
                out_shard.copy_(full[:out_shard.numel()])
            return None
        if self.backend == "nccl":
            return dist.reduce_scatter_tensor(out_shard, full, op=dist.ReduceOp.SUM, group=self.group,
                                              async_op=async_op)
        # gloo: all-reduce then keep the local slice
        work = self.all_reduce_sum(full)
        n = out_shard.numel()
        src = full[self.rank * n:(self.rank + 1) * n]
        if out_shard.data_ptr() != src.data_ptr():
            out_shard.copy_(src)
        return work

    def all_gather(self, full: torch.Tensor, shard: torch.Tensor, async_op: bool = False):
        """``full = concat over ranks of shard``; ``shard`` may alias ``full[rank*n:(rank+1)*n]``."""
        if self.world_size == 1:
            if shard.data_ptr() != full.data_ptr():
                full[:shard.numel()].copy_(shard)
            return None
        if self.backend == "nccl":
            return dist.all_gather_into_tensor(full, shard, group=self.group, async_op=async_op)
        n = shard.numel()
        mine = shard.cpu() if self._staged(shard) else shard.clone()
        chunks: List[torch.Tensor] = [torch.empty_like(mine) for _ in range(self.world_size)]
        dist.all_gather(chunks, mine, group=self.group)
        for r, c in enumerate(chunks):
            full[r * n:(r + 1) * n].copy_(c)
        return None

    def all_reduce_sum(self, t: torch.Tensor, async_op: bool = False):
        if self.world_size == 1:
            return None
        if self._staged(t):
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
            t.copy_(h)
            return None
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)

    def all_reduce_max(self, t: torch.Tensor):
        if self.world_size == 1:
            return None
        if self._staged(t):
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.MAX, group=self.group)
            t.copy_(h)
            return None
        return dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)

    def broadcast(self, t: torch.Tensor, src: int = 0, async_op: bool = False):
        if self.world_size == 1:
            return None
        if self._staged(t):
            h = t.cpu()
            dist.broadcast(h, src=src, group=self.group)
            if self.rank != src:
                t.copy_(h)
            return None
        return dist.broadcast(t, src=src, group=self.group, async_op=async_op)

    def send(self, t: torch.Tensor, dst: int, tag: int = 0):
        return dist.send(t.cpu() if self._staged(t) else t, dst=dst, group=self.group, tag=tag)

    def recv(self, t: torch.Tensor, src: int, tag: int = 0):
        if self._staged(t):
            h = torch.empty(t.shape, dtype=t.dtype)
            r = dist.recv(h, src=src, group=self.group, tag=tag)
            t.copy_(h)
            return r
        return dist.recv(t, src=src, group=self.group, tag=tag)

    def barrier(self):
        if self.world_size > 1:
            if self.backend == "nccl":
                dist.barrier(group=self.group, device_ids=[torch.cuda.current_device()])
            else:
                dist.barrier(group=self.group)
