"""Benchmark of the synthesis hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--dtype f16|bf16|fp32]

One "step" = one pass of `Generator.forward` (feature preparation + HiFi-GAN
vocoder, the scope of the reference's 'generate' timer,
promonet/synthesize/core.py:250-281) over one batch of 32 synthetic 10 s
utterances per GPU (BASELINE.json configs[2]: 861 frames, 220 416 samples
each), inputs resident in HBM, plus - for N > 1 - the all-gather of the
generated audio over xGMI (RCCL). Weights are random-init (no checkpoint is
reachable offline), broadcast from rank 0. Prints ONE JSON line on rank 0.

For N > 1 launch with
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
import argparse
import json
import math
import os
import sys
import time
from pathlib import Path

import torch
import torch.distributed as dist

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

import promonet_amd  # noqa: E402
from promonet_amd import _lib  # noqa: E402

FLOP_PER_SAMPLE = 2_399_772          # conv + convT MACs x 2
ELEMENTS_PER_SAMPLE = 5_101.4        # layer-granular activation elements
PEAK_TFLOPS = {'f16': 2500., 'bf16': 2500., 'fp32': 157.3}   # dense MFMA
PEAK_HBM_GBS = 8000.
# sustained register-resident MFMA rate on random operands under the power cap
# (scripts/micro/mfma_peak.hip, profiles/r01/micro_mfma.txt): what an MFMA
# kernel can reach on this part; reported beside the nominal peak
SUSTAINED_TFLOPS = {'f16': 1650., 'bf16': 1650.}


def parse_args():
    parser = argparse.ArgumentParser()
    parser.add_argument('--gpus', type=int, default=1)
    parser.add_argument('--steps', type=int, default=10)
    parser.add_argument('--warmup', type=int, default=3)
    parser.add_argument('--dtype', default='f16',
                        choices=['f16', 'bf16', 'fp32'])
    parser.add_argument('--batch', type=int, default=32,
                        help='utterances per GPU')
    parser.add_argument('--seconds', type=float, default=10.)
    parser.add_argument('--no-cpu-baseline', action='store_true')
    parser.add_argument('--no-gather', action='store_true')
    return parser.parse_args()


def synthetic_inputs(batch, frames, seed, device):
    """BASELINE.md section 4 synthetic workload, seeded per rank."""
    gen = torch.Generator().manual_seed(seed)
    loudness = torch.rand(batch, 8, frames, generator=gen) * 80. - 100.
    pitch = torch.exp(
        torch.rand(batch, frames, generator=gen) *
        (math.log(550.) - math.log(50.)) + math.log(50.))
    periodicity = torch.rand(batch, frames, generator=gen)
    ppg = torch.softmax(
        3. * torch.randn(batch, 40, frames, generator=gen), dim=1)
    speakers = torch.arange(batch) % promonet_amd.NUM_SPEAKERS
    ones = torch.ones(batch)
    return [t.to(device) for t in (
        loudness, pitch, periodicity, ppg, speakers, ones, ones.clone())]


def cpu_baseline():
    """The CPU oracle (a port of the reference's op sequence in PyTorch
    fp32) timed on this host's cores on a bounded sample of the workload."""
    sys.path.insert(0, str(ROOT / 'oracle'))
    import restatement as oracle
    batch, frames = 4, 430          # 4 x 5 s of audio per run
    state = oracle.random_state(seed=0)
    inputs = oracle.synthetic_inputs(batch, frames, seed=1234)
    samples = batch * frames * promonet_amd.HOPSIZE
    cpus = os.cpu_count() or 1
    default_threads = torch.get_num_threads()
    # oneDNN does not scale to every core on this small a problem: time a few
    # thread counts (bounded: ~10-30 s in total) and report the fastest
    best = None
    tried = {}
    with torch.inference_mode():
        for threads in sorted({min(t, cpus) for t in (16, 32, 64, default_threads)}):
            torch.set_num_threads(threads)
            oracle.generator_forward(*inputs, state)           # warm-up
            times = []
            for _ in range(2):
                start = time.perf_counter()
                oracle.generator_forward(*inputs, state)
                times.append(time.perf_counter() - start)
            seconds = min(times)
            tried[threads] = samples / seconds
            if best is None or seconds < best[1]:
                best = (threads, seconds)
    torch.set_num_threads(default_threads)
    threads, seconds = best
    return {
        'value': samples / seconds, 'unit': 'samples/s', 'cores': threads,
        'kind': 'port',
        'rtf': samples / promonet_amd.SAMPLE_RATE / seconds,
        'samples_per_s_by_threads': tried,
        'sample': f'oracle/restatement.py generator_forward (PyTorch CPU port '
                  f'of the reference op sequence), fp32, batch {batch} x '
                  f'{frames} frames (5 s each), best of 2 after 1 warm-up at '
                  f'the fastest of {sorted(tried)} torch threads on '
                  f'{cpus} cpus'}


def parse_profile(text):
    rows = {}
    for line in text.strip().splitlines():
        label, count, ms, flops, nbytes = line.split()
        rows[label] = {
            'launches': int(count), 'ms': float(ms), 'flops': float(flops),
            'bytes': float(nbytes)}
    return rows


def main():
    args = parse_args()
    rank, world, device = promonet_amd.distributed.init()
    assert world == args.gpus, f'WORLD_SIZE {world} != --gpus {args.gpus}'
    if not torch.cuda.is_available():
        raise RuntimeError('bench.py needs an AMD GPU')

    frames = promonet_amd.convert.seconds_to_frames(args.seconds)
    samples_per_step = args.batch * frames * promonet_amd.HOPSIZE

    promonet_amd.configure(COMPUTE_DTYPE=args.dtype)
    torch.manual_seed(0)
    model = promonet_amd.model.Generator().to(device).eval()
    promonet_amd.distributed.broadcast_model(model)     # RCCL broadcast
    inputs = synthetic_inputs(args.batch, frames, 1234 + rank, device)
    gather = world > 1 and not args.no_gather
    if gather:
        # The all-gather of step k runs on RCCL's stream while step k + 1
        # computes: two destination buffers, one is reused only after its
        # collective has completed (the source is the forward's fresh output
        # tensor, kept alive next to the work handle). Every collective is
        # waited for inside the timed region.
        gathered = [torch.empty(
            world * args.batch, 1, frames * promonet_amd.HOPSIZE,
            device=device) for _ in range(2)]
    pending = [None, None]
    counter = [0]

    def step():
        audio = model(*inputs, None)
        if gather:
            slot = counter[0] & 1
            counter[0] += 1
            if pending[slot] is not None:
                pending[slot][0].wait()
            pending[slot] = (dist.all_gather_into_tensor(
                gathered[slot], audio, async_op=True), audio)
        return audio

    def drain():
        for slot in range(2):
            if pending[slot] is not None:
                pending[slot][0].wait()
                pending[slot] = None

    def fence():
        drain()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    with torch.inference_mode():
        for _ in range(args.warmup):
            step()
        engine = model.model.engine()
        library = _lib.lib()
        library.pm_hifigan_profile_reset(engine)
        library.pm_hifigan_profile_enable(engine, 1)
        fence()
        start = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        elapsed = time.perf_counter() - start
        library.pm_hifigan_profile_enable(engine, 0)
        library.pm_hifigan_profile_collect(engine)
    profile = parse_profile(
        library.pm_hifigan_profile_report(engine).decode())

    if world > 1:
        worst = torch.tensor([elapsed], device=device)
        dist.all_reduce(worst, op=dist.ReduceOp.MAX)
        elapsed = worst.item()

    if rank == 0:
        total_samples = world * samples_per_step * args.steps
        value = total_samples / elapsed
        per_gpu = value / world
        # dominant kernel family (HIP events around every launch, on the
        # launch stream, inside the timed region)
        label, row = max(profile.items(), key=lambda kv: kv[1]['ms'])
        avg_ms = row['ms'] / row['launches']
        flops_per_launch = row['flops'] / row['launches']
        achieved = flops_per_launch / (avg_ms * 1e-3) / 1e12
        traffic = None
        traffic_file = ROOT / 'profiles' / 'traffic.json'
        if traffic_file.exists() and args.batch == 32 and frames == 861:
            traffic = json.loads(traffic_file.read_text()).get(
                f'{label}:{args.dtype}')
        kernel_ms = sum(r['ms'] for r in profile.values()) / args.steps
        result = {
            'metric': f'audio samples/sec (22.05 kHz), batch-{args.batch} '
                      f'{args.seconds:g} s utterances',
            'value': value,
            'unit': 'samples/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3,
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': args.dtype,
            'data': 'synthetic',
            'config': {
                'workload': f'Generator.forward (prepare_features + HiFi-GAN), '
                            f'batch {args.batch} x {args.seconds:g} s per GPU '
                            f'({frames} frames, {frames * 256} samples each), '
                            f'random-init weights, {args.dtype} MFMA operands, '
                            f'fp32 accumulate and activations',
                'batch_per_gpu': args.batch,
                'frames': frames,
                'parallelism': f'batch-sharded x{world}' + (
                    ' + RCCL all-gather of audio (overlapped with the next '
                    'step, drained inside the timed region)' if gather else '')},
            'rtf': value / promonet_amd.SAMPLE_RATE,
            'samples_per_sec_per_gpu': per_gpu,
            'rtf_per_gpu': per_gpu / promonet_amd.SAMPLE_RATE,
            'roofline': {
                'kernel': label,
                'bound': 'mfma',
                'achieved': achieved,
                'peak': PEAK_TFLOPS[args.dtype],
                'unit': 'TFLOP/s',
                'frac': achieved / PEAK_TFLOPS[args.dtype],
                'traffic': traffic,
                'sustained_peak': SUSTAINED_TFLOPS.get(args.dtype),
                'frac_of_sustained_peak': (
                    achieved / SUSTAINED_TFLOPS[args.dtype]
                    if args.dtype in SUSTAINED_TFLOPS else None),
                'avg_launch_ms': avg_ms,
                'launches_per_step': row['launches'] // args.steps,
                'algorithmic_flops_per_launch': flops_per_launch,
                'algorithmic_bytes_per_launch': row['bytes'] / row['launches'],
                'algorithmic_gbs': row['bytes'] / row['launches'] /
                                   (avg_ms * 1e-3) / 1e9,
                'share_of_kernel_time': row['ms'] / sum(
                    r['ms'] for r in profile.values())},
            'whole_path': {
                'tflops': per_gpu * FLOP_PER_SAMPLE / 1e12,
                'frac_of_mfma_peak': per_gpu * FLOP_PER_SAMPLE / 1e12 /
                                     PEAK_TFLOPS[args.dtype],
                'layer_granular_gbs': per_gpu * ELEMENTS_PER_SAMPLE * 4 / 1e9,
                'frac_of_hbm_peak': per_gpu * ELEMENTS_PER_SAMPLE * 4 / 1e9 /
                                    PEAK_HBM_GBS,
                'kernel_ms_per_step': kernel_ms},
            'kernels': {
                k: {'ms_per_step': v['ms'] / args.steps,
                    'launches_per_step': v['launches'] // args.steps,
                    'tflops': v['flops'] / max(v['ms'], 1e-9) / 1e9,
                    'gbs': v['bytes'] / max(v['ms'], 1e-9) / 1e6}
                for k, v in sorted(profile.items())},
        }
        if world == 1 and not args.no_cpu_baseline:
            result['cpu_baseline'] = cpu_baseline()
        print(json.dumps(result))

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
