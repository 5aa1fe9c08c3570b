"""RCCL sanity on the GPU box (one rank, backend nccl): the collectives the
N > 1 path issues - flat broadcast, async all_gather_into_tensor, barrier,
all_reduce(MAX) - run through RCCL itself, not the gloo stand-in of the
single-GPU tests. usage: python scripts/rccl_sanity.py"""
import os
import sys
from pathlib import Path

import torch
import torch.distributed as dist

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import promonet_amd  # noqa: E402

os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29533')
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1)
device = torch.device('cuda:0')
flat = torch.arange(1 << 20, dtype=torch.float32, device=device)
dist.broadcast(flat, src=0)
audio = torch.rand(4, 1, 2560, device=device)
work, out = promonet_amd.distributed.all_gather_into(audio, 1, async_op=True)
work.wait()
assert torch.equal(out, audio)
worst = torch.tensor([1.5, 2.5], device=device)
dist.all_reduce(worst, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
print('rccl ok: backend', dist.get_backend(), 'world', dist.get_world_size(),
      'all_gather', tuple(out.shape), 'max', worst.tolist())
dist.destroy_process_group()
