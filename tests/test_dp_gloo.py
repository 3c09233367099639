"""CPU, world_size 2 (gloo): the data-parallel exchange step `allreduce_grads` averages gradients across ranks
(DP is a NEW capability; ranks hold different batches, one all-reduce per step) -- on a plain module and on
the branch the GPU path takes: `FusedAdamW`'s flat gradient buffers + the `touched` bitmap union, with the two HIP
entry points of the update (`ops.sqnorm_accum`, `ops.adamw_step`) replaced by a torch CPU restatement of the kernels."""
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


def _worker_partial(rank, world, port, out):
    """The plain-module branch with a soft failure on rank 1 only: its second layer never runs (`grad is None`)."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from birdsoundclassif_amd.train import allreduce_grads
    torch.manual_seed(0)
    m = torch.nn.Sequential(torch.nn.Linear(8, 4), torch.nn.Linear(4, 2))
    x = torch.full((3, 8), float(rank + 1))
    h = m[0](x)
    (h.sum() if rank == 1 else m[1](h).sum()).backward()
    had = [p.grad is not None for p in m.parameters()]
    local = [p.grad.clone() if p.grad is not None else torch.zeros_like(p) for p in m.parameters()]
    allreduce_grads(m)
    ok = all(p.grad is not None for p in m.parameters())
    for p, g in zip(m.parameters(), local):
        lst = [torch.zeros_like(g) for _ in range(world)]
        dist.all_gather(lst, g)
        ok = ok and torch.allclose(p.grad, sum(lst) / world, atol=1e-6)
        lst2 = [torch.zeros_like(g) for _ in range(world)]
        dist.all_gather(lst2, p.grad)
        ok = ok and all(torch.equal(lst2[0], t) for t in lst2)
    out[rank] = (bool(ok), had)
    dist.destroy_process_group()


def test_plain_module_branch_with_a_missing_gradient_on_one_rank():
    world = 2
    port = _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker_partial, args=(world, port, out), nprocs=world, join=True)
        res = dict(out)
    assert res[0] == (True, [True, True, True, True]) and res[1] == (True, [True, True, False, False])


def test_allreduce_is_noop_without_process_group():
    from birdsoundclassif_amd.train import allreduce_grads
    m = torch.nn.Linear(3, 2)
    m(torch.ones(1, 3)).sum().backward()
    g = m.weight.grad.clone()
    allreduce_grads(m)
    assert torch.equal(m.weight.grad, g)


# --------------------------------------------------------------------------- FusedAdamW flat-buffer branch
def cpu_sqnorm_accum(g, out):
    """csrc/pointwise_bwd.hip sqnorm_kernel: out (float64[1]) += sum g^2."""
    out += (g.double() ** 2).sum()


def cpu_adamw_step(p, g, m, v, lr, beta1, beta2, eps, wd, step, sqnorm=None, max_norm=0.0):
    """csrc/pointwise_bwd.hip adamw_kernel, same operation order, in fp32."""
    f = torch.float32
    coef = torch.tensor(1.0, dtype=f)
    if sqnorm is not None and max_norm > 0:
        total = sqnorm.sqrt().to(f)
        coef = torch.minimum(torch.tensor(max_norm, dtype=f) / (total + torch.tensor(1e-6, dtype=f)), torch.tensor(1.0, dtype=f))
    bc1 = torch.tensor(1.0 - beta1 ** step, dtype=f)
    bc2s = torch.tensor((1.0 - beta2 ** step) ** 0.5, dtype=f)
    gi = g * coef
    p.mul_(1.0 - lr * wd)
    m.mul_(beta1).add_(gi, alpha=1.0 - beta1)
    v.mul_(beta2).addcmul_(gi, gi, value=1.0 - beta2)
    p.sub_((lr / bc1) * (m / (v.sqrt() / bc2s + eps)))


def patch_cpu_optimizer(monkeypatch=None):
    from birdsoundclassif_amd import ops, train
    from birdsoundclassif_amd.nets import _prep
    set_ = (lambda o, k, v: setattr(o, k, v)) if monkeypatch is None else monkeypatch.setattr
    set_(ops, 'sqnorm_accum', cpu_sqnorm_accum)
    set_(ops, 'adamw_step', cpu_adamw_step)
    set_(train.FusedAdamW, '_check_device', staticmethod(lambda dev: None))


class TwoStage(torch.nn.Module):
    """`first` always gets a gradient; `second` only when the step reaches the second stage (train.py:step)."""

    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.first = torch.nn.Linear(8, 6)
        self.second = torch.nn.Linear(6, 3)
        self.backbone_like = torch.nn.Linear(8, 6)             # second parameter group (lr_backbone)

    def loss(self, x, second_stage):
        hb = self.backbone_like(x)                             # created first, like the backbone's nodes: its backward runs last
        if torch.is_grad_enabled():
            from birdsoundclassif_amd import train
            if train.exchange_armed():                         # what NbmModel._fpn_nhwc does with the backbone's last tap
                hb.register_hook(train.backbone_boundary_hook)
        h = self.first(x) + hb
        l1 = (h ** 2).mean()
        return l1 + (self.second(torch.tanh(h)) ** 2).mean() if second_stage else l1


def _groups(m):
    return [{'params': list(m.first.parameters()) + list(m.second.parameters())},
            {'params': list(m.backbone_like.parameters()), 'lr': 1e-3}]


def _fused_worker(rank, world, port, out, force_rccl_control_flow=False, overlap=False):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    patch_cpu_optimizer()
    from birdsoundclassif_amd import train
    from birdsoundclassif_amd.train import FusedAdamW, allreduce_grads
    train.DP_OVERLAP = overlap
    ctl = None
    if force_rccl_control_flow:
        # the branch a GPU job takes: the default group is RCCL, so the touched bitmap travels through a gloo group of its own,
        # created EAGERLY right after init_process_group (train.__main__ / bench.dist_setup) -- forced here over a gloo default group
        train._device_backend = lambda d: 'nccl'
        ctl = train.init_control_group(dist)
        assert ctl is not None and ctl is not dist.group.WORLD and train._control_group(dist) is ctl
    train.exchange_stats_reset()
    m = TwoStage()
    ref = TwoStage()                                           # single-process torch AdamW on the averaged gradients
    opt = FusedAdamW(_groups(m), lr=1e-2, weight_decay=1e-2)
    opt_ref = torch.optim.AdamW(_groups(ref), lr=1e-2, weight_decay=1e-2)
    ok = True
    # step 0: rank 0 "RPN failed" (first stage only), rank 1 full step; step 1: the other way round; step 2: both fail
    plan = [(False, True), (True, False), (False, False)]
    for it, stages in enumerate(plan):
        x = [torch.full((4, 8), 0.1 * (r + 1) + 0.05 * it) + torch.arange(8.0) * 0.01 for r in range(world)]
        opt.zero_grad()
        train.exchange_begin(opt)
        if it == 2 and rank == 1 and overlap:
            with torch.no_grad():                              # a rank without any backward pass: its hook never fires, the
                m.loss(x[rank], stages[rank])                  # collectives still go out in the same order
        else:
            m.loss(x[rank], stages[rank]).backward()
        allreduce_grads(opt)
        opt.step(max_norm=0.05)
        # reference: average of the per-rank gradients (a rank that skipped the second stage contributes zeros); a
        # parameter that NO rank touched keeps grad None and is skipped, like torch.optim.AdamW does
        opt_ref.zero_grad()
        grads = []
        for r in range(world):
            for p_ in ref.parameters():
                p_.grad = None
            if not (it == 2 and r == 1 and overlap):
                ref.loss(x[r], stages[r]).backward()
            grads.append([None if p_.grad is None else p_.grad.clone() for p_ in ref.parameters()])
        for i, p_ in enumerate(ref.parameters()):
            gs = [g[i] for g in grads if g[i] is not None]
            p_.grad = None if not gs else sum(gs) / world
        torch.nn.utils.clip_grad_norm_([p_ for p_ in ref.parameters() if p_.grad is not None], 0.05)
        opt_ref.step()
        for (n_, a), b in zip(m.named_parameters(), ref.parameters()):
            ok = ok and torch.allclose(a, b, atol=2e-7, rtol=1e-6)
    # replicas identical bit for bit, step counts identical
    for p_ in m.parameters():
        lst = [torch.zeros_like(p_.data) for _ in range(world)]
        dist.all_gather(lst, p_.data.contiguous())
        ok = ok and all(torch.equal(lst[0], t) for t in lst)
    steps = torch.tensor([opt.state[p_]['step'] for p_ in opt.all_params()], dtype=torch.int64)
    lst = [torch.zeros_like(steps) for _ in range(world)]
    dist.all_gather(lst, steps)
    ok = ok and all(torch.equal(lst[0], t) for t in lst)
    # second-stage parameters stepped twice (steps 0 and 1), the others three times
    exp = [3, 3, 2, 2, 3, 3]
    st = train.exchange_stats_summary()
    ok = ok and st is not None and st['steps'] == 3 and st['control_ms'] >= 0 and st['exchange_ms'] >= 0
    if overlap:                                                # the hook started buffer 0's all-reduce inside every backward pass
        ok = ok and st['overlapped_steps'] == (2 if rank == 1 else 3)
    else:
        ok = ok and st['overlapped_steps'] == 0
    out[rank] = bool(ok) and steps.tolist() == exp
    dist.destroy_process_group()


def test_fused_adamw_flat_path_world2_with_a_soft_failure_on_one_rank():
    world = 2
    port = _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_fused_worker, args=(world, port, out), nprocs=world, join=True)
        assert dict(out) == {0: True, 1: True}


def test_rccl_side_control_flow_forced_over_gloo_with_the_overlapped_exchange():
    """the control flow a multi-GPU job takes -- eager gloo control group beside the (here: pretended) RCCL
    default group, touched-bitmap MAX-reduce through it, `set_touched_bitmap`, and the overlapped exchange (buffer 0's all-reduce
    started by the backbone-boundary hook inside the backward pass, buffer 1 after it) -- with a soft failure on one rank per step
    and one step in which rank 1 has no backward pass at all.  Same parameters and step counts on both ranks, equal to a
    single-process AdamW on the averaged gradients."""
    world = 2
    port = _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_fused_worker, args=(world, port, out, True, True), nprocs=world, join=True)
        assert dict(out) == {0: True, 1: True}


def test_fused_adamw_load_state_dict_restores_the_flat_moments(monkeypatch, tmp_path):
    """Checkpoint -> fresh optimizer -> load_state_dict: the kernels read the flat moment buffers, so those must hold
    the loaded moments (and the state must view them); the continued run equals the uninterrupted one."""
    patch_cpu_optimizer(monkeypatch)
    from birdsoundclassif_amd.train import FusedAdamW

    def run(m, opt, its):
        for it in its:
            opt.zero_grad()
            m.loss(torch.full((4, 8), 0.1 + 0.05 * it) + torch.arange(8.0) * 0.01, it % 2 == 0).backward()
            opt.step(max_norm=0.05)

    a = TwoStage()
    oa = FusedAdamW(_groups(a), lr=1e-2, weight_decay=1e-2)
    run(a, oa, range(3))
    torch.save({'model': a.state_dict(), 'optim': oa.state_dict()}, tmp_path / 'c.pt')
    run(a, oa, range(3, 5))

    b = TwoStage()
    ob = FusedAdamW(_groups(b), lr=1e-2, weight_decay=1e-2)
    ck = torch.load(tmp_path / 'c.pt', weights_only=False)
    b.load_state_dict(ck['model'])
    ob.load_state_dict(ck['optim'])
    for f in ob._flat:
        lo, hi = f['m'].data_ptr(), f['m'].data_ptr() + 4 * f['m'].numel()
        for (p_, off, k) in f['spans']:
            st = ob.state[p_]
            assert lo <= st['exp_avg'].data_ptr() < hi and st['exp_avg'].data_ptr() == f['m'][off:].data_ptr()
            assert st['exp_avg_sq'].data_ptr() == f['v'][off:].data_ptr()
            assert p_.data_ptr() == f['p'][off:].data_ptr() and p_.grad.data_ptr() == f['g'][off:].data_ptr()
    assert [ob.state[p_]['step'] for p_ in ob.all_params()] == [3, 3, 2, 2, 3, 3]
    assert float(ob._flat[0]['m'].abs().sum()) > 0
    run(b, ob, range(3, 5))
    for p_, q_ in zip(a.parameters(), b.parameters()):
        assert torch.equal(p_, q_)
    # a later checkpoint of the resumed run stores the LIVE moments
    sd = ob.state_dict()['state']
    assert torch.equal(sd[0]['exp_avg'], ob._flat[0]['m'][:48].view(6, 8))


# --------------------------------------------------------------------------- a rank whose step raises
def _peer_fail_main(rank, port, flavour):
    """Child process: three `train_one_step` calls of a 2-rank job; rank 1's step raises in the third.  Exit code 0 only if all three
    steps returned (i.e. never with the behaviour under test)."""
    import time
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=2)
    patch_cpu_optimizer()
    from birdsoundclassif_amd import train
    train.DP_OVERLAP = True
    if flavour == 'flat':
        train._device_backend = lambda d: 'nccl'               # the GPU job's control flow: bitmap + error bit over their own gloo group
        train.init_control_group(dist)
    m = TwoStage()
    opt = train.FusedAdamW(_groups(m), lr=1e-2, weight_decay=1e-2) if flavour == 'flat' else torch.optim.AdamW(_groups(m), lr=1e-2)

    class Crit:
        weight_dict = {'l': 1.0}

    def fake_step(model, criterion, batch, device, negative_sample, early_backward=False):
        it, x = batch
        if it == 2 and rank == 1:
            raise ValueError('rank 1: broken batch in step 2')
        return {'l': model.loss(x, True)}

    train.step = fake_step
    for it in range(3):
        x = torch.full((4, 8), 0.1 * (rank + 1) + 0.05 * it) + torch.arange(8.0) * 0.01
        t0 = time.perf_counter()
        try:
            train.train_one_step(m, Crit(), opt, (it, x), 0.05, 'cpu', False)
        except Exception as exc:
            print(f'rank {rank} step {it}: {type(exc).__name__}: {exc} [{time.perf_counter() - t0:.2f} s]', flush=True)
            raise SystemExit(3)
        print(f'rank {rank} step {it}: ok', flush=True)
    raise SystemExit(0)


def test_a_rank_whose_step_raises_takes_every_rank_out_of_the_same_step():
    """The error bit behind the touched bitmap (`allreduce_grads(failed=)`): rank 1 raises inside step 2 -> it still joins the step's
    collectives, rank 0 raises PeerStepError in the SAME step; both fresh child processes exit non-zero within seconds instead of
    rank 0 blocking in the all-reduce.  Flat (FusedAdamW, separate control group) and plain-module branch."""
    import subprocess, sys, time
    for flavour in ('flat', 'plain'):
        port = _free_port()
        t0 = time.time()
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), '--peer-fail', str(r), str(port), flavour],
                                  stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                                  cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))) for r in range(2)]
        outs = []
        for p in procs:
            try:
                outs.append(p.communicate(timeout=90)[0])
            except subprocess.TimeoutExpired:
                for q in procs:
                    q.kill()
                raise AssertionError(f'[{flavour}] a rank was still blocked after 90 s')
        assert [p.returncode for p in procs] == [3, 3], outs
        assert 'rank 0 step 0: ok' in outs[0] and 'rank 0 step 1: ok' in outs[0] and 'rank 0 step 2: PeerStepError' in outs[0], outs[0]
        assert 'rank 1 step 1: ok' in outs[1] and 'rank 1 step 2: ValueError: rank 1: broken batch' in outs[1], outs[1]
        assert time.time() - t0 < 60



def test_boundary_hook_leaves_buffer_0_alone_while_an_rpn_share_is_parked(monkeypatch):
    """`parked_flush` back-propagates a parked RPN share into FPN / attention gradients AFTER the backbone-boundary hook has
    fired; the hook must not start buffer 0's all-reduce then (allreduce_grads starts it behind the flush)."""
    from birdsoundclassif_amd import train
    from birdsoundclassif_amd.nets import functional as Fn

    class Opt:
        def flat_grads(self):
            raise AssertionError('the exchange must not start')

    st = {'handles': [], 'started': 0, 't0': None, 'opt': Opt()}
    monkeypatch.setitem(train._PENDING, 1, st)
    monkeypatch.setitem(Fn._PARKED, 123, (None, None, None))
    assert train.backbone_boundary_hook(torch.zeros(1)) is None and st['started'] == 0


if __name__ == '__main__':
    import sys
    if len(sys.argv) == 5 and sys.argv[1] == '--peer-fail':
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        _peer_fail_main(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4])
