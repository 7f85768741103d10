"""World-size-2 test of the head-parallel path on CPU (gloo): every rank owns a
slice of the query heads and of the K/V heads they read, computes attention for
its slice independently (no exchange -- here the oracle stands in for the
kernel, the test is about the sharding and the collective), multiplies with its
slice of the output projection, and ONE all-reduce sums the partial products.
The result must equal the unsharded computation."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import trianglemix_oracle as orc
from vllm_trianglemix_amd.parallel import RowParallelOutputProjection, shard_heads

N, H, HKV, D, HIDDEN = 96, 8, 4, 16, 24
PATTERN = (2, 8, 16)


def _inputs():
    q = orc.bf16_round(orc.seeded_normal(1, (N, H, D)))
    k = orc.bf16_round(orc.seeded_normal(2, (N, HKV, D)))
    v = orc.bf16_round(orc.seeded_normal(3, (N, HKV, D)))
    w = orc.seeded_normal(4, (HIDDEN, H * D)).astype(np.float32)
    return q, k, v, w


def _attend(q, k, v):
    return orc.varlen_gqa_attention(q, lambda b: k, lambda b: v, [0, N], [N], D ** -0.5, "triangle", *PATTERN)


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, kv_heads_total: int, out_path: str, chunk_tokens: int = 0):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        q, k, v, w = _inputs()
        k, v = k[:, :kv_heads_total], v[:, :kv_heads_total]
        sh = shard_heads(H, kv_heads_total, rank, world)
        local = _attend(q[:, sh.head_slice], k[:, sh.kv_head_slice], v[:, sh.kv_head_slice])
        w_shard = torch.from_numpy(np.ascontiguousarray(w[:, sh.head_start * D:(sh.head_start + sh.num_heads) * D]))
        proj = RowParallelOutputProjection(w_shard, chunk_tokens=chunk_tokens)
        y = proj(torch.from_numpy(local.astype(np.float32)))
        if rank == 0:
            np.save(out_path, y.numpy())
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("kv_heads_total", [4, 1])  # partitioned K/V heads; one K/V head shared by both ranks
def test_two_ranks_equal_one(tmp_path, kv_heads_total):
    out_path = str(tmp_path / "y.npy")
    mp.spawn(_worker, args=(2, _free_port(), kv_heads_total, out_path), nprocs=2, join=True)
    q, k, v, w = _inputs()
    k, v = k[:, :kv_heads_total], v[:, :kv_heads_total]
    full = _attend(q, k, v).astype(np.float32).reshape(N, H * D) @ w.T
    got = np.load(out_path)
    np.testing.assert_allclose(got, full, rtol=1e-5, atol=1e-5)


def test_chunked_output_projection_equals_one_all_reduce(tmp_path):
    """chunk_tokens > 0: the token dimension goes through GEMM + all-reduce in pieces (96 tokens in chunks of
    40: two full chunks and a ragged one); the sum must equal the single all-reduce."""
    out_path = str(tmp_path / "y.npy")
    mp.spawn(_worker, args=(2, _free_port(), 4, out_path, 40), nprocs=2, join=True)
    q, k, v, w = _inputs()
    full = _attend(q, k, v).astype(np.float32).reshape(N, H * D) @ w.T
    np.testing.assert_allclose(np.load(out_path), full, rtol=1e-5, atol=1e-5)
