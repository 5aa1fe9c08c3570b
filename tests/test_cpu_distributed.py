"""CPU: the N > 1 path (batch sharding, weight broadcast, audio all-gather)
over gloo with world_size 2. The synthesis function is a stand-in (the HIP
engine needs a GPU); the collectives and the shard arithmetic are the code
that runs over RCCL on the MI355X node."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import promonet_amd
from promonet_amd import distributed


def free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def stand_in(loudness, pitch, periodicity, ppg, speakers, sbr, lr):
    """Deterministic per-utterance 'audio' (B, 1, 256 T)."""
    frame = pitch * sbr[:, None] + periodicity + ppg.sum(1) + \
        loudness.mean(1) * lr[:, None] + speakers[:, None]
    return frame.repeat_interleave(256, dim=-1)[:, None]


class FakeEngineOwner(torch.nn.Module):
    """Stands for HiFiGAN / FARGAN: owns a packed engine built from its
    parameters (here: a flag) that an in-place update must invalidate."""

    def __init__(self):
        super().__init__()
        self.weight = torch.nn.Parameter(torch.rand(3))
        self.packed = True

    def _invalidate(self):
        self.packed = False


class FakeGenerator(torch.nn.Module):

    def __init__(self):
        super().__init__()
        self.model = torch.nn.Sequential(FakeEngineOwner())   # bare container
        self.register_buffer('ppg_threshold', torch.rand(()))
        self._threshold = 0.5


def worker(rank, world, port, total, results):
    os.environ.update(
        RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
        MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    got_rank, got_world, device = distributed.init('gloo')
    assert (got_rank, got_world) == (rank, world) and device.type == 'cpu'

    # weight broadcast: every rank ends with rank 0's tensors
    torch.manual_seed(rank)
    model = torch.nn.Sequential(
        torch.nn.Conv1d(3, 5, 3), torch.nn.Embedding(7, 4))
    model.register_buffer('edges', torch.rand(6))
    distributed.broadcast_model(model)
    torch.manual_seed(0)
    want = torch.nn.Sequential(
        torch.nn.Conv1d(3, 5, 3), torch.nn.Embedding(7, 4))
    want.register_buffer('edges', torch.rand(6))
    for a, b in zip(model.state_dict().values(), want.state_dict().values()):
        assert torch.equal(a, b)

    # engines packed before the broadcast are dropped on every rank, however
    # deep they sit, and host copies of buffers are reset
    fake = FakeGenerator()
    distributed.broadcast_model(fake)
    assert fake.model[0].packed is False and fake._threshold is None

    # sharded synthesis + all-gather (uneven shards when total % world != 0)
    gen = torch.Generator().manual_seed(5)
    frames = 9
    inputs = (
        torch.rand(total, 8, frames, generator=gen),
        torch.rand(total, frames, generator=gen),
        torch.rand(total, frames, generator=gen),
        torch.rand(total, 40, frames, generator=gen),
        torch.arange(total), torch.rand(total, generator=gen),
        torch.rand(total, generator=gen))
    gathered = distributed.synthesize_sharded(stand_in, *inputs)
    assert gathered.shape == (total, 1, frames * 256)
    assert torch.equal(gathered, stand_in(*inputs))
    local = distributed.synthesize_sharded(stand_in, *inputs, gather=False)
    start, end = distributed.shard_bounds(total, rank, world)
    assert local.shape[0] == end - start
    results.put((rank, start, end))
    dist.barrier()
    dist.destroy_process_group()


def run(total):
    context = mp.get_context('spawn')
    results = context.Queue()
    port = free_port()
    processes = [
        context.Process(target=worker, args=(rank, 2, port, total, results))
        for rank in range(2)]
    for process in processes:
        process.start()
    for process in processes:
        process.join(120)
        assert process.exitcode == 0
    bounds = sorted(results.get(timeout=5) for _ in range(2))
    assert bounds[0][1] == 0 and bounds[0][2] == bounds[1][1]
    assert bounds[1][2] == total


def test_world_size_2_even():
    run(total=8)


def test_world_size_2_uneven():
    run(total=5)


def test_world_size_2_fewer_utterances_than_ranks():
    """One utterance, two ranks: rank 1 owns an empty shard, must not call the
    synthesis function (the engine rejects B = 0) and still joins the
    all-gather - the job used to hang here."""
    run(total=1)


def test_shard_bounds_cover_the_batch():
    for total in (1, 7, 32, 256, 257):
        for world in (1, 2, 4, 8):
            pieces = [distributed.shard_bounds(total, r, world)
                      for r in range(world)]
            assert pieces[0][0] == 0 and pieces[-1][1] == total
            for (a, b), (c, d) in zip(pieces, pieces[1:]):
                assert b == c
            sizes = [b - a for a, b in pieces]
            assert max(sizes) - min(sizes) <= 1


def test_single_process_is_a_no_op():
    audio = torch.rand(3, 1, 512)
    assert distributed.all_gather_audio(audio, 3) is audio
    model = torch.nn.Linear(2, 2)
    assert distributed.broadcast_model(model) is model


def test_backend_is_decided_from_the_devices():
    """`distributed.decide_backend`: what carries the weight broadcast and the
    audio all-gather follows from the physical devices the ranks bound."""
    import pytest

    def ranks(ids, host='node'):
        return [{'rank': r, 'host': host, 'index': 0, 'id': i, 'name': 'gpu'}
                for r, i in enumerate(ids)]

    # eight ranks, eight GPUs: RCCL, nothing to report
    chosen, shared, note = distributed.decide_backend(
        ranks([f'pci:00:{b:02x}:00' for b in range(8)]), 8, 8, True)
    assert (chosen, shared, note) == ('nccl', {}, None)
    # every rank sees ONE isolated GPU as cuda:0 (a launcher that masks
    # devices per rank): different identities, still RCCL
    chosen, shared, note = distributed.decide_backend(
        ranks([f'uuid:{b}' for b in range(8)]), 8, 1, True)
    assert chosen == 'nccl' and note is None
    # the same PCI address on two HOSTS is two devices
    entries = ranks(['pci:00:0c:00']) + [dict(ranks(['pci:00:0c:00'], 'other')[0], rank=1)]
    assert distributed.decide_backend(entries, 1, 1, True)[0] == 'nccl'
    # two ranks folded onto the test box's one GPU: gloo, and it says so
    chosen, shared, note = distributed.decide_backend(
        ranks(['pci:00:0c:00'] * 2), 2, 1, True)
    assert chosen == 'gloo' and shared == {('node', 'pci:00:0c:00'): [0, 1]}
    assert 'NOT a scaling measurement' in note and 'ranks [0, 1]' in note
    # a mis-set LOCAL_RANK (two ranks on one of eight GPUs): refused, never a
    # silent host-staged run
    with pytest.raises(RuntimeError, match='LOCAL_RANK'):
        distributed.decide_backend(
            ranks(['pci:00:0c:00'] * 2 + [f'pci:00:{b:02x}:00' for b in range(6)]),
            8, 8, True)
    # forcing: gloo always allowed; nccl on a shared device refused
    assert distributed.decide_backend(
        ranks(['a', 'a']), 2, 1, True, 'gloo') == (
            'gloo', {('node', 'a'): [0, 1]}, None)
    with pytest.raises(RuntimeError, match='RCCL refuses'):
        distributed.decide_backend(ranks(['a', 'a']), 2, 1, True, 'nccl')
    with pytest.raises(ValueError):
        distributed.decide_backend(ranks(['a']), 1, 1, True, 'mpi')
    # CPU ranks (no GPU): gloo, identical 'cpu' identities are not "shared"
    assert distributed.decide_backend(ranks(['cpu'] * 4), 4, 0, False) == (
        'gloo', {}, None)


def caller_group_worker(rank, world, port, results):
    """The process group exists before distributed.init() is ever called (a
    host application that owns torch.distributed): the data plane must take
    ITS backend, not 'nothing initialised' (ADVICE r05)."""
    os.environ.update(
        RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
        MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    assert distributed.backend() == 'gloo'
    assert not distributed._host_staged(torch.zeros(1))   # (CPU tensors: nothing to stage)
    model = torch.nn.Linear(3, 2)
    torch.manual_seed(rank)
    with torch.no_grad():
        model.weight.copy_(torch.rand(2, 3))
    distributed.broadcast_model(model)
    torch.manual_seed(0)
    assert torch.equal(model.weight, torch.rand(2, 3))
    results.put(rank)
    dist.barrier()
    dist.destroy_process_group()


def test_caller_created_group_is_recognised():
    context = mp.get_context('spawn')
    results = context.Queue()
    port = free_port()
    processes = [
        context.Process(target=caller_group_worker, args=(rank, 2, port, results))
        for rank in range(2)]
    for process in processes:
        process.start()
    for process in processes:
        process.join(120)
        assert process.exitcode == 0
    assert sorted(results.get(timeout=5) for _ in range(2)) == [0, 1]
