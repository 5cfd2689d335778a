"""CPU, world_size 2 over gloo: the N>1 path (row sharding + one flat gradient all-reduce)
reproduces the single-process global-batch gradient, and shards are independent of world size."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                "sparse-marginalization-lvm_amd"))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _model():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))


def _local_loss(model, x):
    # normalised by the LOCAL batch like marg.py:126 / structmarg.py:140
    return (torch.softmax(model(x), -1) * torch.arange(3.0)).sum(-1).mean()


def _worker(rank, world, port, out):
    from lvmhelpers import dataparallel as dp
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    x = torch.randn(8, 6, generator=torch.Generator().manual_seed(1))
    model = _model()
    _local_loss(model, dp.shard_rows(x)).backward()
    nbytes = dp.allreduce_gradients(model.parameters())
    init = dp.global_init_scores(8, 4, rank, world, "cpu", seed=42)
    mean = dp.allreduce_mean_scalar(float(rank))
    if rank == 0:
        out.put((torch.cat([p.grad.reshape(-1) for p in model.parameters()]).numpy(), nbytes, mean))
    out.put(("init", rank, init.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_equal_global_batch():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    items = [q.get(timeout=120) for _ in range(3)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    grads = [i for i in items if not isinstance(i[0], str)][0]
    inits = {i[1]: i[2] for i in items if isinstance(i[0], str)}
    x = torch.randn(8, 6, generator=torch.Generator().manual_seed(1))
    model = _model()
    _local_loss(model, x).backward()
    ref = torch.cat([p.grad.reshape(-1) for p in model.parameters()]).numpy()
    assert np.allclose(grads[0], ref, atol=1e-7)
    assert grads[1] == 4 * ref.size and grads[2] == 0.5
    full = torch.rand(8, 8, dtype=torch.double, generator=torch.Generator().manual_seed(42)).numpy()
    assert np.array_equal(np.concatenate([inits[0], inits[1]]), full)


def test_shard_ranges_cover_and_partition():
    from lvmhelpers.dataparallel import shard_range
    for total in (0, 1, 7, 8, 1 << 20, 8193):
        for world in (1, 2, 3, 8):
            r = [shard_range(total, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == total
            assert all(r[k][1] == r[k + 1][0] for k in range(world - 1))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1
