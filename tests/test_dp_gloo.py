"""CPU, world_size 2, gloo: the data-parallel pieces that do not need a GPU --
bucket averaging (what backward() calls per gradient bucket), state broadcast,
batch sharding -- and that the averaged shard gradients equal the gradient of the
mean loss over the global batch (checked with the CPU oracle's classifier head)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from _util import PKG, ROOT, pkg


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import importlib
    dp = importlib.import_module(PKG + ".dp")
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        G = 10  # global batch, uneven split on purpose
        X = torch.randn(G, 32)
        Y = torch.randint(0, 5, (G,))
        W = torch.randn(5, 32, requires_grad=True)
        b, e = dp.shard_range(G, rank, world)
        # replica: mean loss over the local shard, weighted so that the AVERAGE over ranks
        # equals the mean over the global batch (equal shards in bench.py: weight 1)
        wgt = (e - b) * world / G
        loss = torch.nn.functional.cross_entropy(X[b:e] @ W.t(), Y[b:e]) * wgt
        loss.backward()
        red = dp.GradBucketReducer()
        flat = W.grad.detach().clone().flatten()
        red(flat, 1)
        red(None, 0)
        Wr = W.detach().clone().requires_grad_(True)
        torch.nn.functional.cross_entropy(X @ Wr.t(), Y).backward()
        ok = torch.allclose(flat.view_as(Wr.grad), Wr.grad, atol=1e-6)
        # broadcast of model state from rank 0
        lin = torch.nn.Linear(4, 3)
        with torch.no_grad():
            lin.weight.fill_(float(rank + 1))
        dp.broadcast_state(lin, 0)
        ok = ok and bool((lin.weight == 1.0).all()) and red.bytes_reduced == flat.numel() * 4
        ret[rank] = ok
    finally:
        dist.destroy_process_group()


def test_bucket_reducer_world2_gloo():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert all(ret.get(r) for r in range(world)), dict(ret)


def test_shard_range_covers_batch():
    dp = pkg("dp")
    for G in (1, 7, 256, 2048):
        for world in (1, 2, 3, 8):
            spans = [dp.shard_range(G, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == G
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1
