"""GPU: the data-parallel step on REAL ranks over RCCL.  The 2-rank test skips itself on a
1-GPU box; the 1-rank variant runs everywhere and keeps the worker script and the RCCL-next-to-a-replayed-HIP-graph
path exercised."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, 'tests', 'dp_rccl_worker.py')


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _launch(world, extra_env=None):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', PYTHONPATH=ROOT, **(extra_env or {}))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={world}',
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), WORKER]
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)


def test_rccl_world2_dp_gradient_equals_single_gpu_gradient():
    if torch.cuda.device_count() < 2:
        pytest.skip('needs 2 GPUs (the driver\'s 8-GPU node); the 1-rank variant below runs on this box')
    r = _launch(2)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert 'SphereNet world=2' in r.stdout


def test_rccl_single_rank_group_next_to_graph_replay():
    r = _launch(1, {'DIG3D_FORCE_DIST': '1'})
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert 'SphereNet world=1' in r.stdout


def test_world2_engine_on_one_gpu_over_gloo():
    """the real engine under a world of TWO on this 1-GPU box — two processes on device 0, each with its own
    GraphedStep over unequal shards (tiny, config-2-shaped and config-4-shaped), the flat pre-scaled gradient bucket
    all-reduced over gloo (RCCL refuses two ranks on one device; only the transport differs from the 8-GPU job), the
    asynchronous after_replay hook, FlatAdam: the data-parallel gradient equals the single-process gradient of the
    concatenated batch, both replicas end on bit-identical weights; then run.run's data-parallel branch — balanced plan,
    ragged last batch, union pre-capture of the size classes of both ranks before step 0 — tiny and config-4-shaped."""
    r = _launch(2, {'DPW_ONE_GPU': '1'})
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert 'SphereNet world=2' in r.stdout and 'atoms=40-120 hidden=128' in r.stdout and 'atoms=9-29 hidden=128' in r.stdout
    assert 'run.run (config-4-shaped) world=2' in r.stdout
