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
    assert device.type == 'cuda' and dist.get_backend() == 'gloo'

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
         '--warmup', '1', '--batch', '4', '--seconds', '2', '--sustain', '0'],
        capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert done.returncode == 0, done.stderr[-2000:]
    lines = [l for l in done.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    result = json.loads(lines[0])
    assert result['n_gpus'] == 2 and result['world_size'] == 2
    assert result['value'] > 0 and result['scaling'] == 'weak'
