"""CPU, world_size 2 (gloo): the data-parallel exchange step `allreduce_grads` averages gradients across ranks
(DP is a NEW capability; ranks hold different batches, one all-reduce per step)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from birdsoundclassif_amd.train import allreduce_grads
    torch.manual_seed(0)
    m = torch.nn.Sequential(torch.nn.Linear(8, 4), torch.nn.Linear(4, 2))          # same init on every rank
    x = torch.full((3, 8), float(rank + 1))
    m(x).sum().backward()
    local = [p.grad.clone() for p in m.parameters()]
    allreduce_grads(m)
    gathered = [[torch.zeros_like(g) for _ in range(world)] for g in local]
    for g, lst in zip(local, gathered):
        dist.all_gather(lst, g)
    ok = all(torch.allclose(p.grad, sum(lst) / world, atol=1e-6) for p, lst in zip(m.parameters(), gathered))
    # ranks must end up with identical gradients
    same = []
    for p in m.parameters():
        lst = [torch.zeros_like(p.grad) for _ in range(world)]
        dist.all_gather(lst, p.grad)
        same.append(all(torch.equal(lst[0], t) for t in lst))
    out[rank] = bool(ok and all(same))
    dist.destroy_process_group()


def test_allreduce_grads_world2():
    world = 2
    port = _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
        assert dict(out) == {0: True, 1: True}


def test_allreduce_is_noop_without_process_group():
    from birdsoundclassif_amd.train import allreduce_grads
    m = torch.nn.Linear(3, 2)
    m(torch.ones(1, 3)).sum().backward()
    g = m.weight.grad.clone()
    allreduce_grads(m)
    assert torch.equal(m.weight.grad, g)
