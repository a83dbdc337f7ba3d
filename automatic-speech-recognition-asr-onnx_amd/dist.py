"""Utterance-level data parallelism: one process per GPU, full weight replica per GPU.

Utterances (windows) are independent -- the reference has no cross-utterance state outside a
streaming session -- so the path shards with no data-path collective. Two collectives remain
(SURVEY.md section 8e), both through torch.distributed (backend "nccl" == RCCL over xGMI on ROCm,
"gloo" in the CPU tests):
  * start-up: broadcast of the weight arena built by rank 0 (0.47 GB SenseVoice / 3.1 GB Whisper),
  * end of batch: gather of fixed-width hypothesis slabs (B_local, 1 + max_tokens) int32 to rank 0.

Collectives and cluster kernels never share a GPU: every collective issued here on a CUDA device runs inside a FOREIGN SECTION of the native library
(`foreign_section`: asr_device_foreign_begin / _end, include/asr_mi355x.h) -- opening it waits for any SANM block / tile / fused streaming pass in flight on
that GPU, and while it is open every compute call takes its cluster-free path; the section closes only after the collective's kernels have finished
(`torch.cuda.synchronize`). A serving loop that overlaps `gather_hypotheses` of batch k with `run` of batch k + 1 (pool.SessionPool, several threads)
is therefore safe by construction, at the price of that batch's block kernel. Code that issues its own torch.distributed / RCCL calls next to this library
must wrap them the same way.
"""
from __future__ import annotations

import os

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: set before the HIP runtime starts in a multi-process job
from typing import Sequence

import numpy as np
import torch
import torch.distributed as dist


import contextlib


@contextlib.contextmanager
def foreign_section(device):
    """Bracket kernels this library did not launch (RCCL collectives) on `device`: no cluster kernel of the native library runs beside them.
    CPU devices (the gloo tests) pass straight through and never load the native library."""
    dev = torch.device(device)
    if dev.type != "cuda":
        yield
        return
    from . import _lib
    lib = _lib.load()
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    _lib.check(lib.asr_device_foreign_begin(idx))
    try:
        yield
    finally:
        try:
            torch.cuda.synchronize(idx)                  # the collective's kernels (c10d's own stream included) have left the GPU
        finally:
            _lib.check(lib.asr_device_foreign_end(idx))


def foreign_stats(device) -> dict:
    """Counters of the foreign-kernel gate of `device` (process-wide)."""
    import ctypes as C
    from . import _lib
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    out = (C.c_int64 * 4)()
    _lib.check(_lib.load().asr_device_foreign_stats(idx, out))
    return {"sections": int(out[0]), "cluster_passes_diverted": int(out[1]), "sections_that_waited": int(out[2]), "cluster_passes_admitted": int(out[3])}


_GATHER_OK = None        # gather vs send / recv: decided ONCE per process group, by every rank together (a per-call try / except could split the ranks)


def init_from_env(backend: str | None = None):
    """Returns (rank, local_rank, world_size). Initialises the process group when WORLD_SIZE > 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
        global _GATHER_OK
        _GATHER_OK = None                      # a new process group decides again
    return rank, local_rank, world


def shard_bounds(n_items: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous split; the first (n % world) ranks take one extra item."""
    q, r = divmod(n_items, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def shard_by_length(lengths: Sequence[int], world: int) -> list[list[int]]:
    """Length-balanced assignment (longest-first greedy) of utterance indices to ranks."""
    order = sorted(range(len(lengths)), key=lambda i: -int(lengths[i]))
    loads = [0] * world
    out: list[list[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda j: (loads[j], j))
        out[r].append(i)
        loads[r] += int(lengths[i])
    return [sorted(x) for x in out]


def broadcast_arena(blob: np.ndarray | None, device: torch.device, src: int = 0) -> torch.Tensor:
    """Rank `src` passes the host arena; every rank returns a device-resident uint8 tensor holding it."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1:
        return torch.from_numpy(blob).to(device)
    rank = dist.get_rank()
    with foreign_section(device):
        n = torch.zeros(1, dtype=torch.int64, device=device)
        if rank == src:
            n[0] = blob.nbytes
        dist.broadcast(n, src)
        if rank == src:
            t = torch.from_numpy(blob).to(device)
        else:
            t = torch.empty(int(n.item()), dtype=torch.uint8, device=device)
        dist.broadcast(t, src)
    return t


def pack_hypotheses(token_ids: np.ndarray, num_id: np.ndarray, width: int) -> np.ndarray:
    """(B, 1 + width) int32 slab: column 0 = token count, then the ids (zero padded)."""
    B = num_id.shape[0]
    slab = np.zeros((B, 1 + width), dtype=np.int32)
    slab[:, 0] = num_id
    w = min(width, token_ids.shape[1])
    slab[:, 1:1 + w] = token_ids[:, :w]
    return slab


def unpack_hypotheses(slab: np.ndarray) -> list[np.ndarray]:
    return [row[1:1 + row[0]].copy() for row in slab]


def _backend_has_gather(device: torch.device) -> bool:
    """Probe c10d.gather on a one-element tensor; the verdict is all-reduced (MIN) so that every rank takes the same path afterwards."""
    global _GATHER_OK
    if _GATHER_OK is None:
        ok = 1
        with foreign_section(device):
            try:
                t = torch.zeros(1, dtype=torch.int32, device=device)
                bucket = [torch.empty_like(t) for _ in range(dist.get_world_size())] if dist.get_rank() == 0 else None
                dist.gather(t, bucket, dst=0)
            except (RuntimeError, NotImplementedError):
                ok = 0
            v = torch.tensor([ok], dtype=torch.int32, device=device)
            dist.all_reduce(v, op=dist.ReduceOp.MIN)
            _GATHER_OK = bool(int(v.item()))
    return _GATHER_OK


def gather_hypotheses(slab: np.ndarray, device: torch.device, dst: int = 0):
    """All ranks pass equal-shaped slabs; rank `dst` gets the list of per-rank slabs (others None). A true gather (the bytes move
    once, to `dst` only): c10d's gather on both backends (on RCCL it is a group of point-to-point sends), with plain send / recv for a
    backend that lacks it -- which of the two is decided once, collectively (`_backend_has_gather`)."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1:
        return [slab]
    rank = dist.get_rank()
    has_gather = _backend_has_gather(device)
    with foreign_section(device):
        t = torch.from_numpy(np.ascontiguousarray(slab)).to(device)
        bucket = [torch.empty_like(t) for _ in range(world)] if rank == dst else None
        if has_gather:
            dist.gather(t, bucket, dst=dst)
        else:
            if rank == dst:
                bucket[dst].copy_(t)
                for src in range(world):
                    if src != dst:
                        dist.recv(bucket[src], src=src)
            else:
                dist.send(t, dst=dst)
        return [b.cpu().numpy() for b in bucket] if rank == dst else None


def max_over_ranks(value: float, device: torch.device) -> float:
    """The slowest rank's figure (bench.py: elapsed time of the timed region, bracketed by barriers on both sides)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    with foreign_section(device):
        t = torch.tensor([value], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())


def barrier(device) -> None:
    """dist.barrier() inside a foreign section (on RCCL a barrier is an all-reduce kernel)."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        with foreign_section(device):
            dist.barrier()
