"""CPU: bench.py's own N > 1 code - self-launch through torch.distributed.run,
rendezvous on 127.0.0.1, per-rank seeds + weight broadcast, GatherPipeline,
the hang watchdog, the max-over-ranks timing reduction and the `multi_gpu`
block - with EIGHT ranks over gloo and a stand-in model (`--stand-in`: the HIP
engine needs a GPU). What the driver's 8-GPU run executes around the kernels,
made boring before it gets there."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def clean_env():
    env = {k: v for k, v in os.environ.items()
           if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR',
                        'MASTER_PORT', 'LOCAL_WORLD_SIZE')}
    env['OMP_NUM_THREADS'] = '1'
    return env


def run_bench(world, extra=(), env=None, timeout=600):
    return subprocess.run(
        [sys.executable, str(ROOT / 'bench.py'), '--gpus', str(world),
         '--stand-in', '--steps', '3', '--warmup', '1', '--batch', '4',
         '--seconds', '1', '--sustain', '0.2'] + list(extra),
        capture_output=True, text=True, timeout=timeout, env=env or clean_env(),
        cwd=ROOT)


def test_eight_ranks_self_launched():
    done = run_bench(8)
    assert done.returncode == 0, done.stderr[-3000:]
    lines = [l for l in done.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1                       # rank 0 alone prints
    result = json.loads(lines[0])
    frames = 86
    assert result['n_gpus'] == 8 and result['world_size'] == 8
    assert result['steps'] == 3 and result['warmup'] == 1
    assert result['scaling'] == 'weak' and result['higher_is_better'] is True
    assert result['stand_in'] is True and 'stand-in' in result['data']
    assert result['vs_baseline'] is None
    # whole-job aggregate: all ranks' samples over the max-over-ranks time
    samples = 8 * 4 * frames * 256 * 3
    assert result['value'] == pytest.approx(
        samples / (result['ms_per_step'] * 3e-3), rel=1e-9)
    assert result['samples_per_sec_per_gpu'] == pytest.approx(
        result['value'] / 8)
    multi = result['multi_gpu']
    assert multi['world_size_reported_by_backend'] == 8
    assert multi['backend'] == 'gloo' and multi['control_plane_backend'] == 'gloo'
    assert multi['ranks_share_a_device'] is False
    assert multi['gloo_fallback'] is None        # CPU ranks: nothing fell back
    assert len(multi['rank_ms_per_step']) == 8
    assert len(multi['rank_compute_ms_per_step']) == 8
    assert multi['rank_ms_per_step_max'] == pytest.approx(
        result['ms_per_step'], rel=1e-9)
    assert multi['rank_ms_per_step_min'] <= multi['rank_ms_per_step_max']
    assert multi['exposed_gather_ms'] == pytest.approx(
        result['ms_per_step'] - multi['compute_ms_per_step'], abs=1e-9)
    assert multi['gather_alone_ms_per_step'] > 0
    assert multi['gathered_bytes_per_rank_per_step'] == 8 * 4 * frames * 256 * 4
    assert multi['hang_watchdog_seconds'] == 120
    devices = multi['devices']
    assert [d['rank'] for d in devices] == list(range(8))
    assert all(d['id'] == 'cpu' and 'host' in d and 'index' in d
               for d in devices)
    assert result['sustained_steps'] >= 3


def test_under_torchrun_environment():
    """The driver's own command shape: python -m torch.distributed.run ...
    bench.py --gpus N (the environment is given, no self-launch)."""
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    done = subprocess.run(
        [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
         '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
         '--master-port', str(port), str(ROOT / 'bench.py'), '--gpus', '2',
         '--stand-in', '--steps', '2', '--warmup', '1', '--batch', '2',
         '--seconds', '1', '--sustain', '0'],
        capture_output=True, text=True, timeout=600, env=clean_env(), cwd=ROOT)
    assert done.returncode == 0, done.stderr[-3000:]
    lines = [l for l in done.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    result = json.loads(lines[0])
    assert result['n_gpus'] == 2 and 'sustained_ms_per_step' not in result


def test_world_size_mismatch_is_refused():
    """--gpus must agree with the WORLD_SIZE the launcher made."""
    env = clean_env()
    env.update(RANK='0', WORLD_SIZE='1', LOCAL_RANK='0',
               MASTER_ADDR='127.0.0.1', MASTER_PORT='29511')
    done = run_bench(2, env=env)
    assert done.returncode != 0 and 'WORLD_SIZE 1 != --gpus 2' in done.stderr


def test_a_stalled_rank_fails_the_job_fast():
    """One of four ranks never reaches the collectives: the others' watchdogs
    (5 s here) end the job with rc != 0 and say why - no JSON line."""
    import time
    env = clean_env()
    env['PROMONET_BENCH_HANG_SECONDS'] = '5'
    env['PROMONET_BENCH_TEST_STALL_RANK'] = '2'
    begin = time.perf_counter()
    done = run_bench(4, env=env)
    assert done.returncode != 0
    assert time.perf_counter() - begin < 200
    assert 'collective hang' in done.stderr
    assert not [l for l in done.stdout.splitlines() if l.startswith('{')]
