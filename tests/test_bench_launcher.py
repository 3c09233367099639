"""CPU: `python bench.py --gpus N` starts its N ranks itself -- the launcher, the rendezvous and the collective
that proves how many ranks met, with the GPU work stubbed out (NBM_BENCH_DRY=1: gloo, no device).  Also: the same script under
torchrun (the driver's command line), a failing rank, and the refusal to label a smaller job `--gpus N`."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, 'bench.py')


def _run(cmd, **env):
    e = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    e.update(env)
    return subprocess.run(cmd, env=e, capture_output=True, text=True, timeout=240)


def _line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_self_launch_two_ranks():
    r = _run([sys.executable, BENCH, '--gpus', '2', '--steps', '1', '--warmup', '0'], NBM_BENCH_DRY='1')
    assert r.returncode == 0, r.stderr[-2000:]
    d = _line(r.stdout)
    assert d['n_gpus'] == 2 and d['ranks_seen'] == 2 and d['backend'] == 'gloo' and d['rank_devices'] == [0, 1]
    assert d['launcher'] == 'self'
    # N ranks on one host: every rank's host thread pools are capped at cores / N (launcher: environment; rank: torch.set_num_threads),
    # and the control group of the training exchange exists before any step (under gloo it is the default group itself)
    cores = len(os.sched_getaffinity(0))
    assert d['host_threads_per_rank'] == max(1, cores // 2) == d['torch_threads'] and d['OMP_NUM_THREADS'] == str(max(1, cores // 2))
    assert d['control_group'] == 'the default (gloo) group'


def test_same_script_under_torchrun():
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    r = _run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
              '--master-port', str(port), BENCH, '--gpus', '2'], NBM_BENCH_DRY='1')
    assert r.returncode == 0, r.stderr[-2000:]
    d = _line(r.stdout)
    assert d['n_gpus'] == 2 and d['ranks_seen'] == 2 and d['launcher'] == 'torchrun'
    assert d['host_threads_per_rank'] == d['torch_threads'] == max(1, len(os.sched_getaffinity(0)) // 2)


def test_a_failing_rank_fails_the_launch():
    r = _run([sys.executable, BENCH, '--gpus', '2'], NBM_BENCH_DRY='1', NBM_BENCH_DRY_FAIL_RANK='1')
    assert r.returncode != 0
    assert 'rank 1 exited with code 3' in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith('{')]


def test_more_ranks_than_devices_is_refused():
    import torch
    have = torch.cuda.device_count()
    r = _run([sys.executable, BENCH, '--gpus', str(have + 2)])
    assert r.returncode == 2 and 'refusing' in r.stderr and not r.stdout.strip()


def test_gpus_flag_must_match_the_world():
    r = _run([sys.executable, BENCH, '--gpus', '2'], NBM_BENCH_DRY='1', WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
    assert r.returncode != 0 and 'WORLD_SIZE=1' in r.stderr
