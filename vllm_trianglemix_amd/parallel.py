"""Head-parallel sharding of the attention path over the GPUs of one node.

One process per GPU (``torch.distributed``, backend ``nccl`` = RCCL over xGMI
on ROCm, ``gloo`` in CPU tests).  Attention heads are independent, so rank
``r`` of ``p`` owns query heads ``[r*H/p, (r+1)*H/p)`` and the K/V heads they
read -- the rule of the reference's ``Qwen3Attention``
(vllm/model_executor/models/qwen3.py:85-98): K/V heads are partitioned when
there are at least ``p`` of them and replicated otherwise.  The block index is
head independent, so every rank builds the identical index locally and the
attention itself needs no exchange.  The only collective on the path is the
sum of the output projection's partial products
(vllm/model_executor/layers/linear.py:1358-1359), one all-reduce per layer.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch
import torch.distributed as dist


@dataclass(frozen=True)
class HeadShard:
    rank: int
    world: int
    num_heads: int        # query heads on this rank
    num_kv_heads: int     # K/V heads on this rank (>= 1)
    head_start: int       # first global query head
    kv_head_start: int    # first global K/V head
    kv_replicated: bool   # K/V heads are shared with other ranks

    @property
    def head_slice(self) -> slice:
        return slice(self.head_start, self.head_start + self.num_heads)

    @property
    def kv_head_slice(self) -> slice:
        return slice(self.kv_head_start, self.kv_head_start + self.num_kv_heads)


def shard_heads(total_heads: int, total_kv_heads: int, rank: int, world: int) -> HeadShard:
    """Partition of qwen3.py:85-98: ``H % p == 0``; ``Hkv % p == 0`` when
    ``Hkv >= p`` else ``p % Hkv == 0`` and each K/V head serves ``p / Hkv`` ranks."""
    if world <= 0 or not 0 <= rank < world:
        raise ValueError(f"bad rank {rank} of {world}")
    if total_heads % world != 0:
        raise ValueError(f"{total_heads} query heads do not divide over {world} ranks")
    heads = total_heads // world
    if total_kv_heads >= world:
        if total_kv_heads % world != 0:
            raise ValueError(f"{total_kv_heads} K/V heads do not divide over {world} ranks")
        kv = total_kv_heads // world
        return HeadShard(rank, world, heads, kv, rank * heads, rank * kv, False)
    if world % total_kv_heads != 0:
        raise ValueError(f"{world} ranks are not a multiple of {total_kv_heads} K/V heads")
    ranks_per_kv = world // total_kv_heads
    return HeadShard(rank, world, heads, 1, rank * heads, rank // ranks_per_kv, True)


class RowParallelOutputProjection(torch.nn.Module):
    """``o_proj`` of a head-sharded attention: every rank multiplies its local
    heads' outputs with its slice of the weight's input dimension and the
    partial products are summed over ranks with one all-reduce (RCCL over
    xGMI on MI355X).  ``chunk_tokens`` > 0 splits the token dimension so that
    the all-reduce of one chunk overlaps the GEMM of the next on a side stream
    (the all-reduce of a 32k-token prompt is hundreds of MiB)."""

    def __init__(self, weight_shard: torch.Tensor, group: Optional[dist.ProcessGroup] = None,
                 chunk_tokens: int = 0, always_reduce: bool = False):
        super().__init__()
        self.weight_shard = weight_shard  # [hidden, local_heads * head_dim]
        self.group = group
        self.chunk_tokens = chunk_tokens
        # a group of ONE rank normally skips the collective (nothing to sum); with always_reduce the chunked
        # GEMM + all-reduce pipeline runs all the same, so that it can be exercised on a single GPU
        self.always_reduce = always_reduce
        self._comm_stream = None

    def forward(self, attn_out: torch.Tensor) -> torch.Tensor:
        """``attn_out``: ``[T, local_heads, head_dim]`` -> ``[T, hidden]`` (summed over ranks)."""
        t = attn_out.shape[0]
        x = attn_out.reshape(t, -1)
        world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        if world == 1 and not (self.always_reduce and dist.is_initialized()):
            return x @ self.weight_shard.t()
        if self.chunk_tokens <= 0 or t <= self.chunk_tokens:
            y = x @ self.weight_shard.t()
            dist.all_reduce(y, op=dist.ReduceOp.SUM, group=self.group)
            return y
        # chunks of the token dimension: the all-reduce of chunk c runs beside the GEMM of chunk c + 1 (on a
        # second HIP stream for device tensors, as asynchronous collectives for host tensors)
        y = torch.empty((t, self.weight_shard.shape[0]), dtype=x.dtype, device=x.device)
        pending = []
        main = None
        if x.is_cuda:
            if self._comm_stream is None:
                self._comm_stream = torch.cuda.Stream(device=x.device)
            main = torch.cuda.current_stream(x.device)
        for lo in range(0, t, self.chunk_tokens):
            hi = min(t, lo + self.chunk_tokens)
            torch.matmul(x[lo:hi], self.weight_shard.t(), out=y[lo:hi])
            if main is not None:
                self._comm_stream.wait_stream(main)
                with torch.cuda.stream(self._comm_stream):
                    dist.all_reduce(y[lo:hi], op=dist.ReduceOp.SUM, group=self.group)
            else:
                pending.append(dist.all_reduce(y[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        for work in pending:
            work.wait()
        if main is not None:
            main.wait_stream(self._comm_stream)
        return y
