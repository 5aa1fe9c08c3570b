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

import torch
import torch.distributed as dist


def init(backend=None, force=False, timeout=None):
    """Initialise from the torchrun environment (RANK / WORLD_SIZE /
    LOCAL_RANK / MASTER_ADDR / MASTER_PORT). Returns (rank, world, device).
    `force` creates the process group for a single rank too (a 1-GPU box can
    then drive RCCL's collectives, asynchronous overlap included).

    One rank per GPU over RCCL ("nccl"). When there are more local ranks than
    GPUs (a 1-GPU test box running the 2-rank path) the ranks fold onto the
    devices round-robin and the backend falls back to gloo - RCCL refuses two
    ranks on one device - with the collectives staged through host memory.
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
        folded = use_gpu and local_world > devices
        backend = backend or os.environ.get('PROMONET_DIST_BACKEND') or (
            'nccl' if use_gpu and not folded else 'gloo')
        extra = {}
        if timeout is not None:
            import datetime
            extra['timeout'] = datetime.timedelta(seconds=float(timeout))
        dist.init_process_group(backend, rank=rank, world_size=world, **extra)
    return rank, world, device


def _host_staged(tensor):
    """gloo moves host memory: device tensors go through a host copy."""
    return (
        dist.is_initialized() and dist.get_backend() == 'gloo' and
        tensor.is_cuda)


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
            dist.broadcast(host, src=src)
            flat = host.to(flat.device)
        else:
            dist.broadcast(flat, src=src)
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
        dist.all_gather_into_tensor(host, local.cpu())
        out.copy_(host)
        return (None, out) if async_op else out
    work = dist.all_gather_into_tensor(out, local, async_op=async_op)
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
        self.overlap = dist.is_initialized() and dist.get_backend() == 'nccl'
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
