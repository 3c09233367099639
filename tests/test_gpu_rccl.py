"""GPU: RCCL itself, on the one device a test box has.  A one-rank `nccl` process group exercises everything of the data-parallel
exchange that does not need a second GPU: the backend loads and initialises on this ROCm build, `all_reduce` runs on the FusedAdamW
flat gradient buffers with the reduce op the exchange uses (`ReduceOp.AVG`), on int32 (the touched bitmap, `ReduceOp.MAX`) and
float64 (the bench's timing reduction), and `allreduce_grads` itself leaves a one-rank job's gradients untouched.  The multi-rank
semantics are covered on the CPU with gloo (tests/test_dp_gloo.py); RCCL refuses two ranks on one device."""
import os
import socket

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


def test_rccl_one_rank_group_runs_the_collectives_of_the_exchange_step():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{port}', rank=0, world_size=1,
                            device_id=torch.device('cuda', torch.cuda.current_device()))
    try:
        assert dist.get_backend() == 'nccl' and dist.get_world_size() == 1
        flat = torch.arange(1 << 20, device='cuda', dtype=torch.float32)             # a flat gradient buffer
        ref = flat.clone()
        dist.all_reduce(flat, op=dist.ReduceOp.AVG)
        bits = torch.tensor([0, 1, 1, 0, 1], device='cuda', dtype=torch.int32)        # the touched bitmap
        dist.all_reduce(bits, op=dist.ReduceOp.MAX)
        t = torch.tensor([0.125], device='cuda', dtype=torch.float64)                 # bench.py: max-over-ranks time
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        one = torch.ones(1, device='cuda')
        dist.all_reduce(one)                                                          # bench.py: ranks_seen
        dist.barrier()
        torch.cuda.synchronize()
        assert torch.equal(flat, ref) and bits.tolist() == [0, 1, 1, 0, 1] and float(t) == 0.125 and int(one.item()) == 1
        # the touched bitmap of the data-parallel step travels as a HOST tensor through a gloo group created beside the RCCL one
        from birdsoundclassif_amd.train import allreduce_grads, _control_group
        ctl = _control_group(dist)
        assert ctl is not None and _control_group(dist) is ctl                        # created once
        hbits = torch.tensor([0, 1, 1, 0, 1], dtype=torch.int32)
        dist.all_reduce(hbits, op=dist.ReduceOp.MAX, group=ctl)
        assert hbits.tolist() == [0, 1, 1, 0, 1]
        m = torch.nn.Linear(4, 3).cuda()
        m(torch.ones(2, 4, device='cuda')).sum().backward()
        g = m.weight.grad.clone()
        allreduce_grads(m)                                                            # world_size 1: a no-op by contract
        assert torch.equal(m.weight.grad, g)
    finally:
        dist.destroy_process_group()


def _two_ranks(overlap):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    env.update(WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), NBM_DP_OVERLAP='1' if overlap else '0',
               OMP_NUM_THREADS='4', HSA_ENABLE_IPC_MODE_LEGACY='0')
    procs = [subprocess.Popen([sys.executable, os.path.join(root, 'tests', 'dp_gpu_worker.py')], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE if r == 0 else None, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-3000:]
    line = [ln for ln in outs[0][0].splitlines() if ln.startswith('{')]
    assert len(line) == 1, outs[0][0]
    return json.loads(line[0])


def test_two_ranks_share_the_gpu_overlapped_exchange():
    """Two data-parallel ranks of the REAL model on the one GPU of the box (gloo; functional only): the overlapped exchange -- the
    non-backbone buffer's all-reduce started by the hook on the backbone's last tap inside the backward pass -- fires in every step,
    leaves both replicas bit-identical, and averages to the same gradients as the serial exchange (up to the atomic-order noise of
    the weight-gradient kernels between two runs)."""
    import numpy as np
    a = _two_ranks(overlap=True)
    b = _two_ranks(overlap=False)
    assert a['overlap'] is True and b['overlap'] is False
    assert a['stats']['steps'] == 2 and a['stats']['overlapped_steps'] == 2, a['stats']
    assert b['stats']['steps'] == 2 and b['stats']['overlapped_steps'] == 0, b['stats']
    assert a['replicas_identical'] and b['replicas_identical']
    ga, gb = np.array(a['grad_samples']), np.array(b['grad_samples'])
    scale = np.abs(gb).max()
    assert scale > 0 and np.abs(ga - gb).max() < 1e-4 * scale, (np.abs(ga - gb).max(), scale)
    for x, y in zip(a['grad_norms'], b['grad_norms']):
        assert abs(x - y) < 1e-4 * y, (a['grad_norms'], b['grad_norms'])
