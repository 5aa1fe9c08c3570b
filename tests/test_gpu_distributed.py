"""GPU: the N > 1 path with the REAL HIP engine. Two ranks are spawned on the
test box's single GPU (RCCL refuses two ranks on one device, so the ranks
fold onto cuda:0 and the collectives run over gloo, staged through host
memory - `promonet_amd.distributed.init`); what is exercised is everything
else the 8-GPU job runs: weight broadcast into live engines, batch sharding
(even, uneven, fewer utterances than ranks), `Generator.forward` per shard,
the padded fixed-size all-gather. The gathered audio must equal the
unsharded batch BIT FOR BIT (utterances are independent; tile geometry does
not change the summation order)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def worker(rank, world, port, totals, results):
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    sys.path.insert(0, str(root))
    sys.path.insert(0, str(root / 'oracle'))
    os.environ.update(
        RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
        LOCAL_WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
        MASTER_PORT=str(port))
    import torch.distributed as dist
    import promonet_amd
    import restatement as oracle
    from promonet_amd import distributed

    got_rank, got_world, device = distributed.init()
    assert (got_rank, got_world) == (rank, world)
    assert device.type == 'cuda' and distributed.backend() == 'gloo'
    assert distributed.folded() and len(distributed.rank_devices()) == world

    # different weights per rank, engines already built (a warm-up forward)
    # BEFORE the broadcast: non-source ranks must not keep stale packed weights
    torch.manual_seed(100 + rank)
    model = promonet_amd.model.Generator().to(device).eval()
    warm = [t.to(device) for t in oracle.synthetic_inputs(1, 4, seed=1)]
    with torch.inference_mode():
        before = model(*warm, None).clone()
    distributed.broadcast_model(model)
    with torch.inference_mode():
        after = model(*warm, None)
    changed = not torch.equal(before, after)
    assert changed == (rank != 0), 'engine not repacked after the broadcast'

    for total in totals:
        inputs = [t.to(device) for t in
                  oracle.synthetic_inputs(total, 37, seed=7 + total)]
        with torch.inference_mode():
            gathered = distributed.synthesize_sharded(
                lambda *args: model(*args, None), *inputs)
            full = model(*inputs, None)
        assert gathered.shape == full.shape == (total, 1, 37 * 256)
        assert torch.equal(gathered, full), total
        torch.cuda.synchronize()
    results.put((rank, float(full.abs().max())))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_real_engine(device):
    context = mp.get_context('spawn')
    results = context.Queue()
    port = free_port()
    processes = [
        context.Process(
            target=worker, args=(rank, 2, port, (6, 5, 1), results))
        for rank in range(2)]
    for process in processes:
        process.start()
    for process in processes:
        process.join(600)
        assert process.exitcode == 0
    got = sorted(results.get(timeout=5) for _ in range(2))
    # both ranks hold the same gathered audio (same weights after broadcast)
    assert got[0][1] == got[1][1] and got[0][1] > 0


def test_bench_self_launches_two_ranks(device):
    """`python bench.py --gpus 2` with no torchrun environment re-launches
    itself as 2 ranks and prints one JSON line with n_gpus 2 (the driver's
    N = 1 command shape must also work for N > 1)."""
    import json
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    env = {k: v for k, v in os.environ.items()
           if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR',
                        'MASTER_PORT', 'LOCAL_WORLD_SIZE')}
    done = subprocess.run(
        [sys.executable, str(root / 'bench.py'), '--gpus', '2', '--steps', '2',
         '--warmup', '1', '--batch', '4', '--seconds', '2', '--sustain', '0',
         '--no-cpu-baseline'],
        capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert done.returncode == 0, done.stderr[-2000:]
    lines = [l for l in done.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    result = json.loads(lines[0])
    assert result['n_gpus'] == 2 and result['world_size'] == 2
    assert result['value'] > 0 and result['scaling'] == 'weak'
    # the N > 1 line is diagnosable: compute alone, the collective alone, what
    # of it is exposed, every rank's own step time, the backend's own view
    multi = result['multi_gpu']
    assert multi['world_size_reported_by_backend'] == 2
    assert len(multi['rank_ms_per_step']) == 2
    assert multi['rank_ms_per_step_min'] <= multi['rank_ms_per_step_max']
    assert multi['rank_ms_per_step_max'] == pytest.approx(
        result['ms_per_step'], rel=1e-6)
    assert multi['compute_ms_per_step'] > 0
    assert multi['gather_alone_ms_per_step'] > 0
    assert multi['exposed_gather_ms'] == pytest.approx(
        result['ms_per_step'] - multi['compute_ms_per_step'], abs=1e-6)
    assert multi['gathered_bytes_per_rank_per_step'] == 2 * 4 * 172 * 256 * 4
    assert multi['hang_watchdog_seconds'] == 120


def test_bench_eight_folded_ranks_real_engine(device):
    """`python bench.py --gpus 8` on the 1-GPU box: 8 ranks fold onto cuda:0
    and run the REAL engine (batch 4 x 2 s per rank) - eight engines and
    workspaces on one device, eight flat weight broadcasts into packed
    engines (`invalidate_engines`), the sustained-loop step count broadcast
    from rank 0, the fixed-size all-gather through `GatherPipeline` - over
    the gloo data plane the device-identity exchange falls back to, loudly.
    What the 8-GPU job adds to this is RCCL and seven more devices. NOT a
    scaling measurement (no N > 1 curve has been measured: DESIGN.md 7)."""
    import json
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    env = {k: v for k, v in os.environ.items()
           if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR',
                        'MASTER_PORT', 'LOCAL_WORLD_SIZE')}
    done = subprocess.run(
        [sys.executable, str(root / 'bench.py'), '--gpus', '8', '--steps', '2',
         '--warmup', '1', '--batch', '4', '--seconds', '2', '--sustain', '0.2',
         '--no-cpu-baseline'],
        capture_output=True, text=True, timeout=1500, env=env, cwd=root)
    assert done.returncode == 0, done.stderr[-3000:]
    lines = [l for l in done.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    result = json.loads(lines[0])
    assert result['n_gpus'] == 8 and result['world_size'] == 8
    assert result['value'] > 0 and result['scaling'] == 'weak'
    multi = result['multi_gpu']
    assert multi['world_size_reported_by_backend'] == 8
    assert len(multi['rank_ms_per_step']) == 8
    assert len(multi['devices']) == 8
    assert len({entry['id'] for entry in multi['devices']}) == 1   # one GPU
    assert multi['backend'] == 'gloo' and multi['gloo_fallback']
    assert 'gloo' in done.stderr           # (the fallback is announced)
    assert multi['gathered_bytes_per_rank_per_step'] == 8 * 4 * 172 * 256 * 4
    assert multi['compute_ms_per_step'] > 0
    assert multi['gather_alone_ms_per_step'] > 0


def test_bench_fails_fast_on_a_collective_hang(device):
    """A rank that never reaches a collective must fail the job quickly (rc != 0
    well inside the driver's patience), not hang it: rank 1 is made to stall
    in its warm-up (PROMONET_BENCH_TEST_STALL_RANK), rank 0's watchdog -
    shortened to 20 s here - fires inside the first all-gather."""
    import subprocess
    import sys
    import time
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    env = {k: v for k, v in os.environ.items()
           if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR',
                        'MASTER_PORT', 'LOCAL_WORLD_SIZE')}
    env['PROMONET_BENCH_HANG_SECONDS'] = '20'
    env['PROMONET_BENCH_TEST_STALL_RANK'] = '1'
    begin = time.perf_counter()
    done = subprocess.run(
        [sys.executable, str(root / 'bench.py'), '--gpus', '2', '--steps', '2',
         '--warmup', '1', '--batch', '2', '--seconds', '1', '--sustain', '0'],
        capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert done.returncode != 0
    assert time.perf_counter() - begin < 240
    assert 'collective hang' in done.stderr
    assert not [l for l in done.stdout.splitlines() if l.startswith('{')]


###############################################################################
# BASELINE.json configs[3] at its real per-rank size
###############################################################################


def load_bench():
    import importlib.util
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    sys.path.insert(0, str(root))
    spec = importlib.util.spec_from_file_location('bench', root / 'bench.py')
    module = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(module)
    return module


def config4_worker(rank, world, port, backend, results):
    """One rank of the 8-GPU job at its real size: 32 utterances x 861 frames,
    bf16, through the objects bench.py's N > 1 step is made of."""
    os.environ.update(
        RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
        LOCAL_WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
        MASTER_PORT=str(port))
    bench = load_bench()
    import torch.distributed as dist
    import promonet_amd
    from promonet_amd import distributed

    _, _, device = distributed.init(backend=backend, force=True)
    assert distributed.backend() == backend
    assert distributed.folded() == (world > 1)
    batch, frames = 32, 861
    promonet_amd.configure(COMPUTE_DTYPE='bf16')
    torch.manual_seed(rank)                  # broadcast makes them equal
    model = promonet_amd.model.Generator().to(device).eval()
    distributed.broadcast_model(model)
    inputs = bench.synthetic_inputs(batch, frames, 1234 + rank, device)
    pipeline = distributed.GatherPipeline(
        world, (batch, 1, frames * promonet_amd.HOPSIZE), device)
    assert pipeline.overlap == (backend == 'nccl')
    with torch.inference_mode():
        slots = []
        for _ in range(3):                   # both buffers, one reused
            slots.append(pipeline.submit(model(*inputs, None)))
        pipeline.drain()
        torch.cuda.synchronize()
        assert slots == [0, 1, 0]
        gathered = pipeline.result(0)            # the third step's
        assert gathered.shape == (world * batch, 1, frames * 256)
        assert torch.equal(pipeline.result(1), gathered)   # the second's
        # three utterances of the gathered batch (every rank's shard is hit)
        # against their stand-alone synthesis, bit for bit
        gen = torch.Generator().manual_seed(5)
        picks = [int(torch.randint(0, world * batch, (1,), generator=gen))
                 for _ in range(2)] + [world * batch - 1]
        for pick in picks:
            owner, item = divmod(pick, batch)
            theirs = bench.synthetic_inputs(batch, frames, 1234 + owner, device)
            single = model(*[t[item:item + 1] for t in theirs], None)
            assert torch.equal(single[0], gathered[pick]), pick
    results.put((rank, float(gathered.abs().max()), gathered.shape[0]))
    if backend == 'nccl':
        # bench.py's fence(): a device-side barrier on the RCCL data group
        dist.barrier(group=distributed.data_group(), device_ids=[device.index])
        torch.cuda.synchronize()
    dist.barrier()
    distributed.shutdown()


def run_config4(world, backend):
    context = mp.get_context('spawn')
    results = context.Queue()
    port = free_port()
    processes = [
        context.Process(
            target=config4_worker, args=(rank, world, port, backend, results))
        for rank in range(world)]
    for process in processes:
        process.start()
    for process in processes:
        process.join(900)
        assert process.exitcode == 0
    return sorted(results.get(timeout=5) for _ in range(world))


def test_config4_rank_workload_two_ranks(device):
    """Two ranks folded onto the one GPU, each with config 4's per-rank
    workload (32 x 861 frames, 3.6 GB workspace, 2 x 226 MB gather buffers at
    world 8 - here 2 x 56 MB), through bench.py's own step objects; collectives
    over gloo (RCCL refuses two ranks on one device)."""
    got = run_config4(2, 'gloo')
    assert got[0][1] == got[1][1] > 0 and got[0][2] == 64


def test_config4_rccl_overlap_one_rank(device):
    """The asynchronous RCCL branch itself (all_gather_into_tensor on RCCL's
    stream, overlapped with the next forward) with the real 28 MB shard: one
    rank, backend nccl - what a single-GPU box can execute of it."""
    got = run_config4(1, 'nccl')
    assert got[0][1] > 0 and got[0][2] == 32
