"""Multi-GPU synthesis: one process per GPU, batch sharded, RCCL over xGMI.

Utterances are independent (no cross-utterance op in Generator.forward), so
the path shards on the batch axis with replicated weights:
  * one broadcast of the checkpoint tensors from rank 0 at start-up
    (14.2 M parameters, 57 MB fp32) - `broadcast_model`
  * per super-batch, each rank synthesises its contiguous shard and one
    all-gather returns every rank's audio - `synthesize_sharded`
`torch.distributed` backend "nccl" IS RCCL on ROCm; the CPU tests run the
same code over "gloo" with a stand-in synthesis function.
"""
import os
import socket
import sys

import torch
import torch.distributed as dist

# What init() decided. The DEFAULT process group is always gloo: the control
# plane (barriers, timing reductions, the device-identity exchange below) moves
# a few bytes of host memory and must work whatever the GPUs do. The DATA plane
# (weight broadcast, audio all-gather) is a second group over RCCL ("nccl")
# when every rank owns a physical GPU of its own, and the default gloo group -
# device tensors staged through host memory - when ranks share one (a 1-GPU
# test box running the 2-rank path).
_STATE = {
    'group': None,        # data-plane group (None: the default group)
    'backend': None,      # 'nccl' | 'gloo' of the data plane
    'folded': False,      # ranks share a physical device
    'devices': None,      # per rank: {'rank', 'host', 'index', 'id', 'name'}
    'note': None}


def backend():
    """Backend of the data plane: 'nccl' (= RCCL), 'gloo', or None. A process
    group the CALLER created (init() left it alone, _STATE is empty) answers
    for itself: a gloo group must get its device tensors staged through the
    host, an nccl group keeps the gather overlap."""
    if not dist.is_initialized():
        return None
    return _STATE['backend'] or dist.get_backend(_STATE['group'])


def data_group():
    return _STATE['group']


def rank_devices():
    """What every rank runs on (rank, host, device index, PCI identity), as
    exchanged at init - the `multi_gpu.devices` block of bench.py."""
    return _STATE['devices']


def folded():
    return bool(_STATE['folded'])


def device_identity(index):
    """A string that names the PHYSICAL device behind cuda:`index` of this
    process: its UUID / PCI address where torch exposes them, else the
    visibility mask + index (two ranks that each see one isolated GPU as
    cuda:0 must not look like two ranks on one GPU)."""
    props = torch.cuda.get_device_properties(index)
    parts = []
    uuid = getattr(props, 'uuid', None)
    if uuid is not None and set(str(uuid)) - set('0-'):
        parts.append(f'uuid:{uuid}')
    pci = [getattr(props, name, None) for name in (
        'pci_domain_id', 'pci_bus_id', 'pci_device_id')]
    if any(value is not None for value in pci):
        parts.append('pci:' + ':'.join(
            '?' if value is None else f'{value:02x}' for value in pci))
    if not parts:
        mask = ','.join(
            os.environ.get(name, '') for name in (
                'ROCR_VISIBLE_DEVICES', 'HIP_VISIBLE_DEVICES',
                'CUDA_VISIBLE_DEVICES'))
        parts.append(f'visible[{mask}]:{index}')
    return ' '.join(parts)


def decide_backend(entries, local_world, devices, use_gpu, wanted=None):
    """The data plane's backend from what the ranks published (`entries`: one
    {'rank', 'host', 'id', ...} per rank). Returns (backend, shared, note):
    `shared` maps (host, device id) to the ranks that sit on it together,
    `note` is the loud fallback message or None. Pure function of its
    arguments (tests/test_cpu_distributed.py drives every branch).
      all devices different                     -> 'nccl' (RCCL)
      shared, more local ranks than GPUs        -> 'gloo', with a note
      shared, although every rank could own one -> RuntimeError
      `wanted` forces a backend ('nccl' on shared devices -> RuntimeError)"""
    if wanted not in (None, 'nccl', 'gloo'):
        raise ValueError(f'unknown backend {wanted}')
    owners = {}
    for entry in entries:
        owners.setdefault((entry['host'], entry['id']), []).append(
            entry['rank'])
    shared = {k: v for k, v in owners.items() if len(v) > 1} if use_gpu else {}
    note = None
    if shared and wanted != 'gloo':
        text = '; '.join(
            f'ranks {ranks} on {host} share {ident}'
            for (host, ident), ranks in sorted(shared.items()))
        if wanted == 'nccl':
            raise RuntimeError(
                f'promonet_amd.distributed.init: backend nccl asked for but '
                f'{text} (RCCL refuses two ranks on one device)')
        if local_world <= devices:
            raise RuntimeError(
                f'promonet_amd.distributed.init: {text} although {devices} '
                f'GPUs are visible for {local_world} local ranks - LOCAL_RANK '
                'mis-set? Refusing to fall back to host-staged gloo '
                'collectives for a job that could run one rank per GPU over '
                'RCCL.')
        note = (
            f'{text}: {local_world} local ranks on {devices} visible GPU(s) - '
            'data plane over gloo, staged through host memory (test-box mode; '
            'NOT a scaling measurement)')
    chosen = wanted or ('nccl' if use_gpu and not shared else 'gloo')
    return chosen, shared, note


def init(backend=None, force=False, timeout=None):
    """Initialise from the torchrun environment (RANK / WORLD_SIZE /
    LOCAL_RANK / MASTER_ADDR / MASTER_PORT). Returns (rank, world, device).
    `force` creates the process groups for a single rank too (a 1-GPU box can
    then drive RCCL's collectives, asynchronous overlap included).

    One rank per GPU, data plane over RCCL ("nccl"). Which it is gets DECIDED
    FROM THE DEVICES, not from rank arithmetic: every rank publishes the
    physical identity of the GPU it bound (over the gloo control group) and
      * all different -> RCCL;
      * some equal and more local ranks than visible GPUs (LOCAL_WORLD_SIZE >
        device_count: a 1-GPU test box running the 2-rank path) -> the ranks
        stay folded, the data plane falls back to gloo staged through host
        memory, and every rank says so on stderr - such a run exercises the
        plumbing, its timings are NOT a scaling measurement;
      * some equal although every local rank could have had its own GPU (a
        mis-set LOCAL_RANK) -> RuntimeError: a scaling run must not silently
        become a host-staged one.
    `backend` (or PROMONET_DIST_BACKEND) forces the data plane's backend.
    `timeout` (seconds; default: torch's 10 / 30 minutes) bounds every
    collective: a rank that never arrives fails the job instead of hanging it."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    local_world = int(os.environ.get('LOCAL_WORLD_SIZE', str(world)))
    use_gpu = torch.cuda.is_available()
    devices = torch.cuda.device_count() if use_gpu else 0
    index = local % devices if use_gpu else 0
    device = torch.device(f'cuda:{index}' if use_gpu else 'cpu')
    if use_gpu:
        torch.cuda.set_device(device)
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        extra = {}
        if timeout is not None:
            import datetime
            extra['timeout'] = datetime.timedelta(seconds=float(timeout))
        dist.init_process_group('gloo', rank=rank, world_size=world, **extra)
        mine = {
            'rank': rank, 'host': socket.gethostname(), 'index': index,
            'id': device_identity(index) if use_gpu else 'cpu',
            'name': torch.cuda.get_device_name(index) if use_gpu else 'cpu'}
        every = [None] * world
        dist.all_gather_object(every, mine)
        chosen, shared, note = decide_backend(
            every, local_world, devices, use_gpu,
            backend or os.environ.get('PROMONET_DIST_BACKEND'))
        if note:
            sys.stderr.write(
                f'promonet_amd.distributed [rank {rank}]: WARNING: {note}\n')
            sys.stderr.flush()
        group = None
        if chosen == 'nccl':
            group = dist.new_group(backend='nccl', **extra)
        elif chosen != 'gloo':
            raise ValueError(f'unknown backend {chosen}')
        _STATE.update(
            group=group, backend=chosen, folded=bool(shared), devices=every,
            note=note)
    return rank, world, device


def shutdown():
    """Destroy the process groups (data plane first)."""
    if dist.is_initialized():
        dist.destroy_process_group()
    _STATE.update(group=None, backend=None, folded=False, devices=None,
                  note=None)


def _host_staged(tensor):
    """gloo moves host memory: device tensors go through a host copy."""
    return dist.is_initialized() and backend() == 'gloo' and tensor.is_cuda


def shard_bounds(total, rank, world):
    """Contiguous [start, end) of `total` utterances owned by `rank`; the
    first `total % world` ranks take one extra."""
    base, extra = divmod(total, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def broadcast_model(model, src=0):
    """Replicate rank `src`'s parameters and buffers on every rank with one
    flat broadcast per dtype (few, large collectives: xGMI links are
    point-to-point, a ring broadcast is per-link bound). Every HIP engine
    packed from the old tensors is dropped, so the next forward repacks."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return model
    tensors = [t for t in list(model.parameters()) + list(model.buffers())]
    by_dtype = {}
    for tensor in tensors:
        by_dtype.setdefault(tensor.dtype, []).append(tensor)
    for dtype, group in by_dtype.items():
        flat = torch.cat([t.detach().reshape(-1) for t in group])
        if _host_staged(flat):
            host = flat.cpu()
            dist.broadcast(host, src=src, group=data_group())
            flat = host.to(flat.device)
        else:
            dist.broadcast(flat, src=src, group=data_group())
        offset = 0
        with torch.no_grad():
            for tensor in group:
                count = tensor.numel()
                tensor.copy_(flat[offset:offset + count].view_as(tensor))
                offset += count
    invalidate_engines(model)
    return model


def invalidate_engines(model):
    """In-place parameter updates bypass `_apply` and the load_state_dict
    hooks: drop every packed engine (HiFiGAN / FARGAN, bare or inside a
    Generator) and every host-side copy of a buffer."""
    for module in model.modules():
        if hasattr(module, '_invalidate'):
            module._invalidate()
        elif hasattr(module, '_destroy'):
            module._destroy()
        if hasattr(module, '_threshold'):
            module._threshold = None


def all_gather_audio(local, total, world=None):
    """All-gather variable-size shards (B_r, 1, S) into (total, 1, S).

    Shards differ by at most one utterance; they are padded to the largest
    shard so a single fixed-size all-gather (one large collective) suffices.
    """
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world = world or dist.get_world_size()
    largest = -(-total // world)
    padded = local
    if local.shape[0] < largest:
        pad = torch.zeros(
            (largest - local.shape[0],) + tuple(local.shape[1:]),
            dtype=local.dtype, device=local.device)
        padded = torch.cat((local, pad))
    gathered = all_gather_into(padded.contiguous(), world)
    pieces = []
    for rank in range(world):
        start, end = shard_bounds(total, rank, world)
        pieces.append(
            gathered[rank * largest:rank * largest + (end - start)])
    return torch.cat(pieces)


def all_gather_into(local, world=None, out=None, async_op=False):
    """Fixed-size all-gather of `local` (n, ...) into (world * n, ...).
    Returns the gathered tensor, or (work, tensor) when async_op (RCCL only:
    the collective then runs on RCCL's stream beside the caller's kernels)."""
    world = world or dist.get_world_size()
    if out is None:
        out = torch.empty(
            (world * local.shape[0],) + tuple(local.shape[1:]),
            dtype=local.dtype, device=local.device)
    if _host_staged(local):
        host = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(host, local.cpu(), group=data_group())
        out.copy_(host)
        return (None, out) if async_op else out
    work = dist.all_gather_into_tensor(
        out, local, group=data_group(), async_op=async_op)
    return (work, out) if async_op else out


class GatherPipeline:
    """The all-gather of step k's audio overlapped with the synthesis of step
    k + 1: two destination buffers, a buffer is reused only after its
    collective has completed. Over RCCL the collective is asynchronous (it
    runs on RCCL's stream beside the caller's kernels; the source tensor is
    kept alive next to the work handle); over gloo (ranks folded onto fewer
    GPUs) it is synchronous and staged through host memory. `bench.py`'s step
    for N > 1 and the config-4 test drive exactly this object."""

    def __init__(self, world, shard_shape, device, dtype=torch.float32):
        self.world = world
        self.overlap = dist.is_initialized() and backend() == 'nccl'
        self.buffers = [
            torch.empty(
                (world * shard_shape[0],) + tuple(shard_shape[1:]),
                dtype=dtype, device=device) for _ in range(2)]
        self.pending = [None, None]
        self.count = 0

    def submit(self, audio):
        """Start gathering `audio` (this rank's shard); returns the slot."""
        slot = self.count & 1
        self.count += 1
        self.wait(slot)
        work, _ = all_gather_into(
            audio, self.world, out=self.buffers[slot], async_op=True)
        self.pending[slot] = (work, audio)
        return slot

    def wait(self, slot):
        if self.pending[slot] is not None:
            if self.pending[slot][0] is not None:
                self.pending[slot][0].wait()
            self.pending[slot] = None

    def result(self, slot):
        """The gathered (world * B, ...) tensor of `slot`, complete."""
        self.wait(slot)
        return self.buffers[slot]

    def drain(self):
        for slot in range(2):
            self.wait(slot)


def synthesize_sharded(
    synthesize, loudness, pitch, periodicity, ppg, speakers,
    spectral_balance_ratios, loudness_ratios, gather=True, hopsize=None,
    dtype=torch.float32
):
    """Run `synthesize` on this rank's shard of a (replicated) global batch.

    `synthesize(loudness, pitch, periodicity, ppg, speakers, sbr, lr)` ->
    (B_r, 1, S), e.g. `promonet_amd.model.Generator.forward`. Returns the
    gathered (B, 1, S) audio on every rank (or the local shard). With fewer
    utterances than ranks the surplus ranks synthesise nothing (the engine
    rejects empty batches) and contribute an empty shard to the collective:
    `hopsize` (default: the configured promonet_amd.HOPSIZE) and `dtype` give
    that shard the shape and type the other ranks' audio has.
    """
    if hopsize is None:
        import promonet_amd
        hopsize = promonet_amd.HOPSIZE
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    total = pitch.shape[0]
    start, end = shard_bounds(total, rank, world)
    if end > start:
        local = synthesize(
            loudness[start:end], pitch[start:end], periodicity[start:end],
            ppg[start:end], speakers[start:end],
            spectral_balance_ratios[start:end], loudness_ratios[start:end])
    else:
        local = torch.zeros(
            0, 1, pitch.shape[-1] * hopsize, dtype=dtype,
            device=pitch.device)
    return all_gather_audio(local, total, world) if gather else local
