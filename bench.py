"""Benchmark of the synthesis hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
                    [--model hifigan|fargan] [--dtype f16|bf16|fp32]
                    [--batch B] [--seconds S]

One "step" = one pass of `Generator.forward` (feature preparation + vocoder,
the scope of the reference's 'generate' timer,
promonet/synthesize/core.py:250-281) over one batch of 32 synthetic 10 s
utterances per GPU (BASELINE.json configs[2]: 861 frames, 220 416 samples
each), inputs resident in HBM, plus - for N > 1 - the all-gather of the
generated audio over xGMI (RCCL). Weights are random-init (no checkpoint is
reachable offline), broadcast from rank 0. Prints ONE JSON line on rank 0.

`--gpus N` with N > 1 and no torchrun environment re-launches this script as
N ranks (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N
--master-addr 127.0.0.1 ...`); under torchrun it uses the environment it is
given. Other BASELINE.json configs: `--dtype fp32 --batch 8 --seconds 5`
(config 2), `--model fargan` (config 5).

At N = 1 the line also carries a `secondary` block, measured in the same run
after the headline region (about 20 s of GPU time; `--no-secondary` skips it):
the library-default 'checkpoint' operand mode on the headline workload, config
2 (fp32, 8 x 5 s), config 5 (FARGAN, 'mixed' and 'fp32' weight storage), the
three preprocess transforms through the C ABI, and the latency of the
graph-replayed `packed_inference` at nn~ chunk sizes - each with its time, its
throughput and (where one kernel dominates) that kernel's roofline fraction.

`--stand-in` (tests/test_cpu_bench_multi_rank.py) replaces the HIP engine by a
trivial torch function so that the N > 1 harness itself - self-launch,
rendezvous, weight broadcast, GatherPipeline, watchdog, max-over-ranks
reduction, the `multi_gpu` block - runs on a host without GPUs over gloo. Its
JSON line says so and is not a measurement.
"""
import argparse
import json
import math
import os
import socket
import statistics
import subprocess
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
FARGAN_FLOP_PER_SAMPLE = 73_843
# FARGAN weights read per dependent sub-frame step (fargan.py:199-335): framewise
# conv 256x520 + its GLU 256x256, 3 x (GRU 768x384 + 768x256 + GLU 256x256),
# skip 256x1152 + GLU, output 64x256; + 1/4 of the per-frame conditioning net
FARGAN_STEP_WEIGHTS = 2_246_656
# (with f16 storage only: the fp32 / mixed modes run the conditioning network
# for every frame as three GEMMs ahead of the recurrence, pm_fargan_cond_kernel)
FARGAN_COND_WEIGHTS = 2 * 371 * 371 + 512 * 371
# of those: 3 x (768 x 384 + 768 x 256) GRU + 5 x 256 x 256 GLU gate weights
FARGAN_MIXED_F16_WEIGHTS = 3 * 768 * 640 + 5 * 256 * 256
# dense MFMA peaks per USEFUL flop (f16x3: three MFMAs per product)
PEAK_TFLOPS = {'f16': 2500., 'bf16': 2500., 'fp32': 157.3,
               'f16x3': 2500. / 3, 'f16a2': 2500. / 2}
PEAK_HBM_GBS = 8000.
# (the SUSTAINED rate of the matrix pipe under this device's power cap is
# measured in the run: mfma_probe below)


def parse_args():
    parser = argparse.ArgumentParser()
    parser.add_argument('--gpus', type=int, default=1)
    parser.add_argument('--steps', type=int, default=10)
    parser.add_argument('--warmup', type=int, default=3)
    parser.add_argument('--model', default='hifigan',
                        choices=['hifigan', 'fargan'])
    parser.add_argument('--dtype', default=None,
                        help='MFMA operand type f16|bf16|fp32|f16x3, or one per '
                             'upsampling stage joined by + (hifigan, default '
                             'bf16; f16+f16+f16+f16x3 = the trained-checkpoint '
                             'mode, DESIGN.md section 3) / '
                             'stored weight type f16|mixed|fp32 (fargan, default '
                             'fp32; its math is always fp32)')
    parser.add_argument('--batch', type=int, default=32,
                        help='utterances per GPU')
    parser.add_argument('--seconds', type=float, default=10.)
    parser.add_argument('--sustain', type=float, default=5.,
                        help='seconds of back-to-back steps timed after the '
                             'K official ones (steady-state clocks); 0 = off')
    parser.add_argument('--no-cpu-baseline', action='store_true')
    parser.add_argument('--no-traffic', action='store_true',
                        help='skip the two rocprofv3 PMC passes that measure '
                             'the HBM traffic of every kernel (roofline.traffic '
                             'then comes from profiles/traffic.json)')
    parser.add_argument('--no-gather', action='store_true')
    parser.add_argument('--no-secondary', action='store_true',
                        help='skip the secondary block (checkpoint mode, '
                             'config 2, FARGAN, preprocess, latency)')
    parser.add_argument('--stand-in', action='store_true',
                        help='plumbing test of the N > 1 harness: a trivial '
                             'torch function instead of the HIP engine (runs '
                             'without a GPU, over gloo); not a measurement')
    return parser.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` outside torchrun: become the launcher."""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    command = [
        sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
        f'--nproc-per-node={args.gpus}', '--master-addr', '127.0.0.1',
        '--master-port', str(port), str(Path(__file__).resolve())
    ] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    raise SystemExit(subprocess.run(command, env=env).returncode)


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


def cpu_baseline(model_name, budget=200.):
    """The CPU oracle (a port of the reference's op sequence in PyTorch fp32)
    timed on this host's cores as SURVEY.md section 8(d) specifies:
      * config 2 (batch 8 x 430 frames; FARGAN: batch 2): one warm-up run, then
        the MEDIAN OF 3, at 8 torch threads and at min(32, cpus) - the two
        settings that have been fastest on every host of the pool (all-core
        settings only document oversubscription: 64 / 128 threads measured
        0.6-0.8x of 32 in rounds 2-3);
      * config 3 (batch 32 x 861 frames = 7 053 312 samples; FARGAN: batch 8)
        ONCE at the faster setting, in chunks of 8 utterances (the host-memory
        footprint of config 2);
      * config 1 (one 2 s `from_features`-sized call), median of 3.
    About 100-130 s of CPU work on the pool's hosts. On a host so slow that the
    warm-up run alone says the plan cannot fit `budget` seconds the runs are
    cut (single run instead of a median, config 3 skipped) and the entry says
    so. A reported baseline, not the target."""
    sys.path.insert(0, str(ROOT / 'oracle'))
    import restatement as oracle
    if model_name == 'fargan':
        c2_batch, c2_frames, c3_batch, chunk = 2, 430, 8, 2
        state = oracle.random_state_fargan(seed=0)
        forward = oracle.fargan_generator_forward
        name = 'fargan_generator_forward'
    else:
        c2_batch, c2_frames, c3_batch, chunk = 8, 430, 32, 8
        state = oracle.random_state(seed=0)
        forward = oracle.generator_forward
        name = 'generator_forward'
    c3_frames = 861
    hop = promonet_amd.HOPSIZE
    cpus = os.cpu_count() or 1
    default_threads = torch.get_num_threads()
    settings = sorted({min(8, cpus), min(32, cpus)})

    def run(batch, frames, chunk_size=None):
        inputs = oracle.synthetic_inputs(batch, frames, seed=1234)
        chunk_size = chunk_size or batch
        start = time.perf_counter()
        for first in range(0, batch, chunk_size):
            forward(*[t[first:first + chunk_size] for t in inputs], state)
        return time.perf_counter() - start

    c2_samples = c2_batch * c2_frames * hop
    c3_samples = c3_batch * c3_frames * hop
    by_threads, sample_by_threads = {}, {}
    begin = time.perf_counter()
    spent = lambda: time.perf_counter() - begin     # noqa: E731
    with torch.inference_mode():
        for threads in settings:
            torch.set_num_threads(threads)
            warm = run(c2_batch, c2_frames)                       # warm-up
            # the plan from here: 3 runs now, 4 more at the other setting,
            # config 3 (8 x config 2): ~15 runs' worth
            if spent() + 3 * warm > .6 * budget:
                by_threads[threads] = c2_samples / warm
                sample_by_threads[threads] = (
                    f'{c2_batch} x {c2_frames} frames (config 2), the warm-up '
                    f'run only ({warm:.0f} s: slow host)')
                continue
            times = [run(c2_batch, c2_frames) for _ in range(3)]
            by_threads[threads] = c2_samples / statistics.median(times)
            sample_by_threads[threads] = (
                f'{c2_batch} x {c2_frames} frames (config 2), median of 3 '
                'after a warm-up')
        threads = max(by_threads, key=by_threads.get)
        torch.set_num_threads(threads)
        # config 1: one 2 s utterance through the public call's op sequence
        short_frames = promonet_amd.convert.seconds_to_frames(2.)
        run(1, short_frames)
        short = statistics.median([run(1, short_frames) for _ in range(3)])
        # config 3 itself, once
        predicted = c3_samples / by_threads[threads]
        if spent() + predicted <= 1.25 * budget:
            seconds = run(c3_batch, c3_frames, chunk)
            config3 = {
                'seconds': seconds, 'threads': threads,
                'samples_per_s': c3_samples / seconds,
                'rtf': c3_samples / seconds / promonet_amd.SAMPLE_RATE,
                'sample': f'{c3_batch} x {c3_frames} frames, one run in '
                          f'chunks of {chunk} utterances'}
        else:
            config3 = {
                'skipped': f'predicted {predicted:.0f} s does not fit the '
                           f'{budget:.0f} s budget on this host'}
    total = spent()
    torch.set_num_threads(default_threads)
    return {
        'value': by_threads[threads], 'unit': 'samples/s', 'cores': threads,
        'kind': 'port',
        'rtf': by_threads[threads] / promonet_amd.SAMPLE_RATE,
        'samples_per_s_by_threads': by_threads,
        'sample_by_threads': sample_by_threads,
        'config1_2s_utterance': {
            'seconds': short, 'threads': threads,
            'samples_per_s': short_frames * hop / short,
            'rtf': short_frames * hop / short / promonet_amd.SAMPLE_RATE,
            'sample': f'1 x {short_frames} frames, median of 3 after a warm-up'},
        'config3': config3,
        'host_cpus': cpus,
        'sample': f'oracle/restatement.py {name} (PyTorch CPU port of the '
                  f'reference op sequence), fp32: config 2 ({c2_batch} x '
                  f'{c2_frames} frames) as warm-up + median of 3 at '
                  f'{sorted(by_threads)} torch threads of {cpus} cpus, config 1 '
                  f'(1 x {short_frames} frames) and config 3 ({c3_batch} x '
                  f'{c3_frames} frames, once) at the faster setting ({threads}); '
                  f'{total:.0f} s of CPU work; value = config 2 at the faster '
                  'setting'}


class Watchdog:
    """N > 1: a collective that hangs (a rank that died, a fabric fault) must
    fail the job - rc 3 within `seconds` - instead of stalling the driver. Armed
    around every phase that contains a collective; torch's own collective
    timeout (distributed.init(timeout=...)) is the second line."""

    def __init__(self, seconds=120.):
        self.seconds, self.timer = seconds, None

    def arm(self, what, seconds=None):
        import threading
        self.disarm()
        seconds = seconds or self.seconds

        def fire():
            sys.stderr.write(
                f'bench.py: rank {os.environ.get("RANK", "0")}: "{what}" did '
                f'not finish within {seconds:.0f} s - collective hang? '
                'aborting\n')
            sys.stderr.flush()
            os._exit(3)
        self.timer = threading.Timer(seconds, fire)
        self.timer.daemon = True
        self.timer.start()

    def disarm(self):
        if self.timer is not None:
            self.timer.cancel()
            self.timer = None


def operand_type_of(label, dtype):
    """The MFMA operand type a kernel label (`block_c128_k11`, `convT_c64_r2`,
    `mrf_c32` ...) ran with when `dtype` names one type per stage."""
    import re
    parts = dtype.split('+')
    if parts == ['checkpoint']:
        from promonet_amd.model.hifigan import checkpoint_schedule
        parts = checkpoint_schedule(4)

    def resolve(name, upsampler):
        # a stage's type names its Blocks; 'f16a2' / 'f16ux' stages run their
        # upsampler fully split (promonet_hip.h)
        if name in ('f16a2', 'f16ux'):
            return 'f16x3' if upsampler else {'f16ux': 'f16'}.get(name, name)
        return name
    if len(parts) == 1:
        return resolve(parts[0], label.startswith('convT'))
    match = re.search(r'_c(\d+)', label)
    if not match:
        return resolve(parts[0], False)
    channels = int(match.group(1))
    initial = promonet_amd.HIFIGAN_UPSAMPLE_INITIAL_SIZE
    for stage in range(len(parts)):
        # upsamplers are labelled with their input width, Blocks with the output's
        width = initial >> stage if label.startswith('convT') \
            else initial >> (stage + 1)
        if width == channels:
            return resolve(parts[stage], label.startswith('convT'))
    return resolve(parts[0], False)


def kernel_label(name):
    """rocprofv3 kernel name -> the label the engine's profile report uses."""
    import re
    name = name.replace(' ', '').replace('Elem', '')
    m = re.search(r'conv_pair_kernel<\w+,(\d+),(\d+),', name)
    if m:
        return f'pair_c{m.group(1)}_k{m.group(2)}'
    m = re.search(r'conv_block3(?:_walk|_skew)?_kernel<\w+,(\d+),(\d+),', name)
    if m:
        return f'block_c{m.group(1)}_k{m.group(2)}'
    m = re.search(r'conv_mrf(?:_walk|_skew)?_kernel<\w+,(\d+),', name)
    if m:
        return f'mrf_c{m.group(1)}'
    if 'pm_out_conv' in name:
        return 'out_conv_tanh'
    if re.search(r'conv_single_kernel<\w+,7,7,', name):
        return 'input_conv'
    m = re.search(r'conv_upsample_kernel<\w+,(\d+),', name)
    if m:
        return f'convT_c{m.group(1)}_r8'
    if re.search(r'conv_single_kernel<\w+,2,3,64,4,2,1,\d,0>', name):
        return 'convT_c128_r2'
    if re.search(r'conv_single_kernel<\w+,2,3,64,2,2,1,\d,0>', name):
        return 'convT_c64_r2'
    if 'pm_fargan_cluster_kernel' in name:
        return 'fargan_cluster'
    if 'pm_fargan_cond_kernel' in name:
        return 'fargan_cond'
    return None


def measure_traffic(args, timeout=150.):
    """HBM bytes per launch of every kernel of the step, MEASURED in this run:
    two rocprofv3 PMC passes (FETCH_SIZE, then WRITE_SIZE - they do not fit one
    pass: MI355X_MICROARCH.md, rocprofv3 PMC slots) over one step of this very
    command. bytes = (FETCH_SIZE x 2 + WRITE_SIZE) x 1024: both counters are in
    KiB and on gfx950 FETCH_SIZE reports half of a wide coalesced read stream
    (the guide's HBM section). Returns ({label: bytes}, note) or (None, why)."""
    import csv
    import shutil
    import tempfile
    tool = shutil.which('rocprofv3') or '/opt/rocm/bin/rocprofv3'
    if not Path(tool).exists():
        return None, 'rocprofv3 not found'
    command = [
        sys.executable, str(Path(__file__).resolve()), '--steps', '1',
        '--warmup', '1', '--sustain', '0', '--no-cpu-baseline', '--no-traffic',
        '--no-secondary',
        '--model', args.model, '--dtype', args.dtype, '--batch',
        str(args.batch), '--seconds', str(args.seconds)]
    env = dict(os.environ, TMPDIR='/tmp')
    sums = {}
    with tempfile.TemporaryDirectory(dir='/tmp') as scratch:
        for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
            out = Path(scratch) / counter
            try:
                done = subprocess.run(
                    [tool, '--output-format', 'csv', '--pmc', counter,
                     '--kernel-trace', '-d', str(out), '-o', counter, '--'] +
                    command, cwd='/tmp', env=env, capture_output=True,
                    text=True, timeout=timeout)
            except subprocess.TimeoutExpired:
                return None, f'rocprofv3 {counter} pass timed out'
            files = list(out.rglob('*counter_collection.csv'))
            if done.returncode != 0 or not files:
                return None, f'rocprofv3 {counter} pass failed'
            values, counts = {}, {}
            for file in files:
                for row in csv.DictReader(open(file)):
                    if row.get('Counter_Name') != counter:
                        continue
                    label = kernel_label(row['Kernel_Name'])
                    if label is None:
                        continue
                    values[label] = values.get(label, 0.) + \
                        float(row['Counter_Value'])
                    counts[label] = counts.get(label, 0) + 1
            # (the warm-up step and the profiled step: average per dispatch)
            sums[counter] = {k: values[k] / counts[k] for k in values}
    table = {
        label: (sums['FETCH_SIZE'][label] * 2 +
                sums['WRITE_SIZE'].get(label, 0.)) * 1024
        for label in sums['FETCH_SIZE']}
    return table, ('measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE '
                   '(separate passes over one step of the same command), '
                   '(FETCH_SIZE x 2 + WRITE_SIZE) x 1024 per dispatch')


def parse_profile(text):
    rows = {}
    for line in text.strip().splitlines():
        label, count, ms, flops, nbytes = line.split()
        rows[label] = {
            'launches': int(count), 'ms': float(ms), 'flops': float(flops),
            'bytes': float(nbytes)}
    return rows


class StandIn(torch.nn.Module):
    """`--stand-in`: what the multi-rank harness needs of a Generator - a
    parameter to broadcast, a forward that returns (B, 1, 256 T) audio."""

    def __init__(self):
        super().__init__()
        self.weight = torch.nn.Parameter(torch.rand(8))

    def forward(self, loudness, pitch, periodicity, ppg, speakers, sbr, lr,
                previous=None):
        frame = (pitch * 1e-3 * sbr[:, None] + periodicity + ppg.sum(1) +
                 loudness.mean(1) * 1e-2 * lr[:, None] +
                 speakers[:, None] * self.weight.sum())
        return frame.repeat_interleave(promonet_amd.HOPSIZE, dim=-1)[:, None]


def sync(device):
    if device.type == 'cuda':
        torch.cuda.synchronize(device)


def time_events(fn, reps, warmup=3):
    """Average milliseconds of `fn` over `reps` back-to-back calls, HIP events
    on the launch stream (torch's current stream is the one handed to the C
    ABI)."""
    for _ in range(warmup):
        fn()
    begin = torch.cuda.Event(enable_timing=True)
    end = torch.cuda.Event(enable_timing=True)
    begin.record()
    for _ in range(reps):
        fn()
    end.record()
    end.synchronize()
    return begin.elapsed_time(end) / reps


def mfma_probe(operand, device, target_ms=50.):
    """The matrix pipe's sustained rate on this device, now: a register-resident
    loop of v_mfma_f32_32x32x16 (pm_mfma_probe), sized to ~`target_ms`, timed
    with HIP events. Returns TFLOP/s or None (operand type without a probe)."""
    code = {'f16': _lib.PM_F16, 'bf16': _lib.PM_BF16, 'f16x3': _lib.PM_F16,
            'f16a2': _lib.PM_F16}.get(operand)
    if code is None:
        return None
    library = _lib.lib()
    dtype = torch.bfloat16 if code == _lib.PM_BF16 else torch.float16
    operands = (torch.rand(32768, device=device) * 2 - 1).to(dtype)
    sink = torch.zeros(1, device=device)
    cus = torch.cuda.get_device_properties(device).multi_processor_count
    grid = 2 * cus

    def run(iterations):
        begin = torch.cuda.Event(enable_timing=True)
        end = torch.cuda.Event(enable_timing=True)
        begin.record()
        _lib.check(library.pm_mfma_probe(
            code, iterations, operands.data_ptr(), sink.data_ptr(), grid,
            _lib.stream()))
        end.record()
        end.synchronize()
        return begin.elapsed_time(end)
    run(200)
    iterations = 2000
    ms = run(iterations)
    iterations = max(200, int(iterations * target_ms / max(ms, 1e-3)))
    ms = run(iterations)
    return grid * 4. * iterations * 16 * 32768 / (ms * 1e-3) / 1e12


def fargan_roofline(storage, batch, frames, avg_ms, per_gpu_samples_per_s):
    """FARGAN's `roofline` object from one forward's HIP-event time. What bounds
    this model is not HBM (0.2 GB of compulsory traffic per launch) nor the
    matrix pipe (74 kFLOP per sample) but the memory system one level up:
    every cluster member re-streams its eighth of the weights from its XCD's
    L2 on each of the 3 444 dependent steps, between 6 inter-workgroup
    exchanges per step. `achieved` / `peak` are therefore the aggregate L2 ->
    CU weight stream against the L2's measured 34.5 TB/s
    (MI355X_MICROARCH.md); the per-step latency of the recurrence stands
    beside it (its phase breakdown is a profile, not a bench number:
    profiles/r04/fargan/timeline_fargan.txt)."""
    steps = frames * 4
    # ('mixed': the GRU cells and the GLU gates - 1 802 240 of the weights a
    # sub-frame step streams - are stored f16, the rest fp32)
    wbytes = {
        'f16': (FARGAN_STEP_WEIGHTS + FARGAN_COND_WEIGHTS // 4) * 2,
        'mixed': FARGAN_STEP_WEIGHTS * 4 - FARGAN_MIXED_F16_WEIGHTS * 2,
    }.get(storage, FARGAN_STEP_WEIGHTS * 4)
    # a one-utterance cluster member keeps its seven short slices in the LDS
    # (FgResident, pm_fargan.h: sizes from its static_asserts): not streamed
    if batch <= 32:
        wbytes -= 8 * {'mixed': 122_880, 'f16': 102_400}.get(storage, 106_496)
    # compulsory HBM bytes of one launch: features in, audio out, the weights
    # once (they stay L2-resident for all 3 444 steps)
    hbm_bytes = batch * frames * (128 * 4 + 256 * 4) + wbytes
    clusters = min(32, batch)
    l2_gbs = wbytes * steps * clusters / (avg_ms * 1e-3) / 1e9
    return {
        'kernel': 'pm_fargan_cluster_kernel',
        # (not an HBM fraction: `achieved` / `peak` / `frac` are the L2 -> CU
        # weight stream; the recurrence is latency-bound, see latency_model)
        'bound': 'l2', 'level': 'l2 (weights re-streamed per step)',
        'achieved': l2_gbs, 'peak': 34500., 'unit': 'GB/s',
        'frac': l2_gbs / 34500., 'traffic': None,
        'avg_launch_ms': avg_ms,
        'algorithmic_bytes_per_launch': hbm_bytes,
        'hbm_gbs': hbm_bytes / (avg_ms * 1e-3) / 1e9,
        'note': 'latency-bound recurrence: see latency_model; HBM itself '
                'carries only hbm_gbs',
        'latency_model': {
            'dependent_steps': steps,
            'us_per_step': avg_ms * 1e3 / steps,
            'per_cu_l2_stream_gbs': l2_gbs / (clusters * 8),
            'per_cu_l2_peak_gbs': 34500. / 256,
            'exchanges_per_step': 6,
            'matrix_slices_per_step': 13,
            'tflops': per_gpu_samples_per_s * FARGAN_FLOP_PER_SAMPLE / 1e12}}


###############################################################################
# Secondary block: every other single-GPU claim, measured in the driver's run
###############################################################################


def secondary_hifigan(dtype, batch, seconds, device, steps=6, warmup=2):
    """One HiFi-GAN configuration: K forwards timed with HIP events, then a
    per-kernel pass for the dominant kernel and its MFMA fraction."""
    library = _lib.lib()
    previous = promonet_amd.COMPUTE_DTYPE
    promonet_amd.configure(MODEL='hifigan', COMPUTE_DTYPE=dtype)
    try:
        torch.manual_seed(0)
        model = promonet_amd.model.Generator().to(device).eval()
        frames = promonet_amd.convert.seconds_to_frames(seconds)
        inputs = synthetic_inputs(batch, frames, 1234, device)
        with torch.inference_mode():
            ms = time_events(lambda: model(*inputs, None), steps, warmup)
            engine = model.model.engine()
            library.pm_hifigan_profile_only(engine, None)
            library.pm_hifigan_profile_reset(engine)
            library.pm_hifigan_profile_enable(engine, 1)
            table_steps = 3
            for _ in range(table_steps):
                model(*inputs, None)
            library.pm_hifigan_profile_enable(engine, 0)
            library.pm_hifigan_profile_collect(engine)
            table = parse_profile(
                library.pm_hifigan_profile_report(engine).decode())
        label, row = max(table.items(), key=lambda kv: kv[1]['ms'])
        operand = operand_type_of(label, dtype)
        launch_ms = row['ms'] / row['launches']
        tflops = row['flops'] / row['launches'] / (launch_ms * 1e-3) / 1e12
        samples = batch * frames * promonet_amd.HOPSIZE
        value = samples / (ms * 1e-3)
        stem = operand_type_of('input_conv', dtype)
        return {
            'workload': f'Generator.forward (prepare_features + HiFi-GAN), '
                        f'batch {batch} x {seconds:g} s ({frames} frames), '
                        f'{dtype} MFMA operands, fp32 accumulate',
            'dtype': dtype, 'batch': batch, 'frames': frames,
            'steps': steps, 'warmup': warmup,
            'ms_per_step': ms, 'samples_per_s': value,
            'rtf': value / promonet_amd.SAMPLE_RATE,
            'whole_path_tflops': value * FLOP_PER_SAMPLE / 1e12,
            'whole_path_frac_of_mfma_peak':
                value * FLOP_PER_SAMPLE / 1e12 / PEAK_TFLOPS[stem],
            'kernel_ms_per_step':
                sum(r['ms'] for r in table.values()) / table_steps,
            'dominant_kernel': {
                'kernel': label, 'operands': operand,
                'avg_launch_ms': launch_ms,
                'launches_per_step': row['launches'] // table_steps,
                'ms_per_step': row['ms'] / table_steps,
                'achieved': tflops, 'peak': PEAK_TFLOPS[operand],
                'unit': 'TFLOP/s', 'bound': 'mfma',
                'frac': tflops / PEAK_TFLOPS[operand]}}
    finally:
        promonet_amd.configure(COMPUTE_DTYPE=previous)


def secondary_fargan(storage, device, batch=32, seconds=10., steps=2, warmup=1):
    """BASELINE.json configs[4]: config/fargan.py, batch 32 x 10 s."""
    default = promonet_amd.FARGAN_WEIGHT_DTYPE
    promonet_amd.configure(MODEL='fargan', FARGAN_WEIGHT_DTYPE=storage)
    try:
        torch.manual_seed(0)
        model = promonet_amd.model.Generator().to(device).eval()
        frames = promonet_amd.convert.seconds_to_frames(seconds)
        inputs = synthetic_inputs(batch, frames, 1234, device)
        with torch.inference_mode():
            ms = time_events(lambda: model(*inputs, None), steps, warmup)
        samples = batch * frames * promonet_amd.HOPSIZE
        value = samples / (ms * 1e-3)
        return {
            'workload': f'Generator.forward (prepare_features + FARGAN, '
                        f'config/fargan.py), batch {batch} x {seconds:g} s '
                        f'({frames} frames = {frames * 4} dependent sub-frame '
                        f'steps), weights stored as {storage}, fp32 arithmetic',
            'weight_storage': storage,
            'library_default_storage': storage == default,
            'dtype': 'fp32', 'steps': steps, 'warmup': warmup,
            'ms_per_step': ms, 'samples_per_s': value,
            'rtf': value / promonet_amd.SAMPLE_RATE,
            'dominant_kernel': fargan_roofline(
                storage, batch, frames, ms, value)}
    finally:
        promonet_amd.configure(MODEL='hifigan', FARGAN_WEIGHT_DTYPE=default)


def secondary_preprocess(device, batch=32, seconds=10., reps=400):
    """spectrogram.from_audio, from_audio(mels=True) and loudness.from_audio
    (8 bands) of `batch` x `seconds` of audio through the C ABI on
    preallocated buffers (spectrogram.py:15-60,111-133; loudness.py:17-55):
    HBM-bound streaming kernels, algorithmic bytes = 4 B / sample in (twice
    for the two loudness passes) + 4 B per output value."""
    library = _lib.lib()
    frames = promonet_amd.convert.seconds_to_frames(seconds)
    samples = frames * promonet_amd.HOPSIZE
    bins = promonet_amd.NUM_FFT // 2 + 1
    mels = promonet_amd.NUM_MELS
    gen = torch.Generator().manual_seed(1234)
    audio = (torch.randn(batch, samples, generator=gen) * .1).to(device)
    spec = torch.empty(batch, bins, frames, device=device)
    mel = torch.empty(batch, mels, frames, device=device)
    loud = torch.empty(batch, 8, frames, device=device)
    weights = promonet_amd.preprocess.loudness.perceptual_weights_tensor(device)
    prepared = promonet_amd.preprocess.spectrogram._prepared_mel_basis(device)
    scratch = torch.empty(
        max(1, library.pm_loudness_scratch_bytes(batch, samples)),
        dtype=torch.uint8, device=device)
    stream = _lib.stream()
    calls = {
        'spectrogram': (lambda: _lib.check(library.pm_stft_magnitude(
            _lib.ptr(audio), _lib.ptr(spec), batch, samples, None, 0, stream)),
            4. * batch * samples + 4. * spec.numel(), 1),
        'log_mel': (lambda: _lib.check(library.pm_stft_mel(
            _lib.ptr(audio), prepared.data_ptr(), _lib.ptr(mel), batch,
            samples, mels, 0, 0., stream)),
            4. * batch * samples + 4. * mel.numel(), 1),
        'loudness_8_bands': (lambda: _lib.check(library.pm_loudness(
            _lib.ptr(audio), _lib.ptr(weights), _lib.ptr(loud), batch, samples,
            8, float(promonet_amd.MIN_DB), scratch.data_ptr(), scratch.numel(),
            stream)),
            2 * 4. * batch * samples + 4. * loud.numel(), 2)}
    out = {
        'workload': f'{batch} x {seconds:g} s of audio ({frames} frames each) '
                    'through the C ABI, buffers preallocated',
        'dtype': 'fp32', 'reps': reps}
    for name, (call, nbytes, launches) in calls.items():
        # (hundreds of back-to-back launches after a long warm-up: a burst of
        # 50 of these 40 us kernels reads 10 % high - the clocks are still
        # settling from the model builds in front of it)
        ms = time_events(call, reps, warmup=100)
        gbs = nbytes / (ms * 1e-3) / 1e9
        out[name] = {
            'ms': ms, 'kernel_launches': launches,
            'audio_seconds_per_second': batch * seconds / (ms * 1e-3),
            'algorithmic_bytes': nbytes, 'bound': 'hbm',
            'achieved': gbs, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
            'frac': gbs / PEAK_HBM_GBS}
    return out


def secondary_latency(device, chunks=(8, 16, 32, 64), reps=200):
    """The streaming (nn~) use of the generator, generator.py:334-343:
    `packed_inference` on one (1, 53, T) buffer, eager and as a replayed
    hipGraph (`graph=True`), host wall-clock per call including the sync."""
    promonet_amd.configure(COMPUTE_DTYPE=promonet_amd.DEFAULT_COMPUTE_DTYPE)
    torch.manual_seed(0)
    model = promonet_amd.model.Generator().to(device).eval()
    out = {'workload': 'Generator.packed_inference, batch 1, '
                       f'{promonet_amd.DEFAULT_COMPUTE_DTYPE} operand mode, host '
                       'wall-clock per call (submit + synchronize), median',
           'reps': reps}
    gen = torch.Generator().manual_seed(7)
    with torch.inference_mode():
        for frames in chunks:
            packed = torch.rand(1, 53, frames, generator=gen).to(device)
            row = {}
            for name, kwargs in (('eager', {}), ('graph', {'graph': True})):
                for _ in range(5):
                    model.packed_inference(packed, **kwargs)
                torch.cuda.synchronize()
                times = []
                for _ in range(reps):
                    begin = time.perf_counter()
                    model.packed_inference(packed, **kwargs)
                    torch.cuda.synchronize()
                    times.append(time.perf_counter() - begin)
                row[f'{name}_us'] = statistics.median(times) * 1e6
            chunk_seconds = frames * promonet_amd.HOPSIZE / promonet_amd.SAMPLE_RATE
            row['chunk_ms_of_audio'] = chunk_seconds * 1e3
            row['graph_rtf'] = chunk_seconds / (row['graph_us'] * 1e-6)
            out[f'frames_{frames}'] = row
    return out


def secondary_block(device, budget=45.):
    """Every single-GPU claim besides the headline, on this run's clock. An
    entry that fails reports its error instead of taking the line down; an
    entry that would start past `budget` seconds is skipped and says so."""
    plan = [
        ('hifigan_checkpoint_batch32_10s',
         lambda: secondary_hifigan('checkpoint', 32, 10., device)),
        ('hifigan_fp32_config2_batch8_5s',
         lambda: secondary_hifigan('fp32', 8, 5., device)),
        ('fargan_mixed_batch32_10s', lambda: secondary_fargan('mixed', device)),
        ('fargan_fp32_batch32_10s', lambda: secondary_fargan('fp32', device)),
        ('preprocess_batch32_10s', lambda: secondary_preprocess(device)),
        ('packed_inference_latency', lambda: secondary_latency(device))]
    out = {}
    begin = time.perf_counter()
    for name, run in plan:
        spent = time.perf_counter() - begin
        if spent > budget:
            out[name] = {'skipped': f'{spent:.0f} s of the {budget:.0f} s '
                                    'secondary budget already spent'}
            continue
        try:
            out[name] = run()
        except Exception as error:        # noqa: BLE001 (reported, not fatal)
            out[name] = {'error': f'{type(error).__name__}: {error}'}
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
    out['seconds'] = time.perf_counter() - begin
    return out


def main():
    args = parse_args()
    if args.dtype is None:
        # (bf16: what BASELINE.json's batch-32 x 10 s config names; the
        # library's own default operand mode, 'checkpoint', is in `secondary`)
        # (fargan: arithmetic is fp32 in every mode; the flag is the STORAGE of
        # the streamed weights and defaults to the library's default, 'fp32';
        # --dtype mixed: GRU cells / GLU gates stored f16, 6.5e-6 max-abs over
        # a whole 10 s utterance. Both are in the hifigan line's `secondary`)
        args.dtype = promonet_amd.FARGAN_WEIGHT_DTYPE if args.model == 'fargan' \
            else 'bf16'
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        self_launch(args)
    watchdog = Watchdog(float(os.environ.get('PROMONET_BENCH_HANG_SECONDS', 120)))
    if args.gpus > 1:
        # (ranks reach the rendezvous up to a minute or two apart on a fresh
        # box - the first `import torch` pages the image in)
        watchdog.arm('process-group initialisation', 300.)
    rank, world, device = promonet_amd.distributed.init(
        timeout=300. if args.gpus > 1 else None)
    if world != args.gpus:
        raise SystemExit(
            f'bench.py: WORLD_SIZE {world} != --gpus {args.gpus}')
    stand_in = args.stand_in
    if not torch.cuda.is_available() and not stand_in:
        raise RuntimeError('bench.py needs an AMD GPU')
    fargan = args.model == 'fargan' and not stand_in
    hifigan = not fargan and not stand_in

    frames = promonet_amd.convert.seconds_to_frames(args.seconds)
    samples_per_step = args.batch * frames * promonet_amd.HOPSIZE

    if fargan:
        weight_dtype = args.dtype if args.dtype in ('f16', 'mixed') else 'fp32'
        promonet_amd.configure(MODEL='fargan', FARGAN_WEIGHT_DTYPE=weight_dtype)
    elif hifigan:
        promonet_amd.configure(COMPUTE_DTYPE=args.dtype)
    # (every rank starts from its OWN weights: what makes them equal is the
    # broadcast below, as in the job that loads a checkpoint on rank 0)
    torch.manual_seed(rank if world > 1 else 0)
    model = (StandIn() if stand_in else promonet_amd.model.Generator()).to(
        device).eval()
    if world > 1:
        watchdog.arm('weight broadcast')
    promonet_amd.distributed.broadcast_model(model)     # RCCL broadcast
    inputs = synthetic_inputs(args.batch, frames, 1234 + rank, device)
    gather = world > 1 and not args.no_gather
    backend = promonet_amd.distributed.backend() if world > 1 else None
    # N > 1: the all-gather of step k runs on RCCL's stream while step k + 1
    # computes (promonet_amd.distributed.GatherPipeline: two destination
    # buffers, every collective waited for inside the timed region)
    pipeline = promonet_amd.distributed.GatherPipeline(
        world, (args.batch, 1, frames * promonet_amd.HOPSIZE), device) \
        if gather else None
    overlap = gather and pipeline.overlap

    def step():
        audio = model(*inputs, None)
        if gather:
            pipeline.submit(audio)
        return audio

    def drain():
        if gather:
            pipeline.drain()

    def fence():
        drain()
        sync(device)
        if world > 1:
            # (over RCCL when it carries the data plane: a device-side barrier
            # of tens of microseconds inside the timed region instead of a
            # host-side gloo one of up to a millisecond at 8 ranks)
            if backend == 'nccl':
                dist.barrier(group=promonet_amd.distributed.data_group(),
                             device_ids=[device.index])
            else:
                dist.barrier()
            sync(device)

    library = None if stand_in else _lib.lib()
    engine = None
    forward_events = []
    def guarded(what):
        if world > 1:
            watchdog.arm(what)

    if os.environ.get('PROMONET_BENCH_TEST_STALL_RANK') == str(rank) and world > 1:
        # test hook (tests/test_gpu_distributed.py): this rank never reaches
        # the collectives - the others' watchdogs must end the job
        watchdog.disarm()
        time.sleep(3600)
    with torch.inference_mode():
        guarded('warm-up steps')
        for _ in range(args.warmup):
            step()
        kernel_table = None
        if hifigan:
            # Per-kernel table: a separate pass of K steps with HIP events
            # around EVERY launch (not part of `value`: 34 event pairs per step
            # cost 0.7 % of it). The timed region then brackets only the
            # dominant kernel's launches - what `roofline` is computed from.
            engine = model.model.engine()
            fence()
            guarded('per-kernel pass')
            library.pm_hifigan_profile_only(engine, None)
            library.pm_hifigan_profile_reset(engine)
            library.pm_hifigan_profile_enable(engine, 1)
            table_steps = min(args.steps, 5)
            for _ in range(table_steps):
                audio = model(*inputs, None)
            library.pm_hifigan_profile_enable(engine, 0)
            library.pm_hifigan_profile_collect(engine)
            kernel_table = parse_profile(
                library.pm_hifigan_profile_report(engine).decode())
            dominant = max(kernel_table.items(), key=lambda kv: kv[1]['ms'])[0]
            library.pm_hifigan_profile_reset(engine)
            library.pm_hifigan_profile_only(engine, dominant.encode())
            library.pm_hifigan_profile_enable(engine, 1)
        fence()
        guarded('timed region')
        start = time.perf_counter()
        for _ in range(args.steps):
            if fargan:
                # HIP events on the launch stream (torch's current stream is
                # the one the C ABI is handed) around the forward's launches
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                step()
                e1.record()
                forward_events.append((e0, e1))
            else:
                step()
        fence()
        elapsed = time.perf_counter() - start
        if hifigan:
            library.pm_hifigan_profile_enable(engine, 0)
            library.pm_hifigan_profile_collect(engine)
            timed_profile = parse_profile(
                library.pm_hifigan_profile_report(engine).decode())
            library.pm_hifigan_profile_only(engine, None)
        # steady state: the official region can be shorter than the DVFS /
        # power-cap settling time, so also time >= `sustain` seconds of
        # back-to-back steps (reported beside, never instead of, `value`)
        sustained = None
        if args.sustain > 0:
            per_step = max(elapsed / args.steps, 1e-4)
            count = max(args.steps, int(math.ceil(args.sustain / per_step)))
            count = min(count, int(60. / per_step) + 1)
            if world > 1:
                # every rank must run the SAME number of steps (each one is a
                # collective): rank 0's count, not one from each rank's clock
                agreed = torch.tensor([count], dtype=torch.int64)
                dist.broadcast(agreed, src=0)
                count = int(agreed.item())
            fence()
            guarded('sustained loop')
            begin = time.perf_counter()
            for _ in range(count):
                step()
            fence()
            sustained = (time.perf_counter() - begin, count)

        # N > 1: what a sub-linear scaling number would be made of - the same
        # K steps WITHOUT the collective (compute alone, per rank) and the
        # collective alone (submitted and drained, nothing to hide under)
        diagnosis = None
        if world > 1:
            guarded('compute-only steps')
            fence()
            begin = time.perf_counter()
            for _ in range(args.steps):
                audio = model(*inputs, None)
            sync(device)
            compute_own = time.perf_counter() - begin
            fence()
            gather_alone = 0.
            if gather:
                guarded('gather-only steps')
                begin = time.perf_counter()
                for _ in range(args.steps):
                    pipeline.submit(audio)
                    pipeline.drain()
                fence()
                gather_alone = time.perf_counter() - begin
            diagnosis = (compute_own, gather_alone)

    if world > 1:
        watchdog.arm('timing reduction')
        # (host doubles over the gloo control group)
        own = torch.tensor(
            [elapsed, sustained[0] if sustained else 0., diagnosis[0],
             diagnosis[1]], dtype=torch.float64)
        every = [torch.empty_like(own) for _ in range(world)]
        dist.all_gather(every, own)
        every = torch.stack(every)                            # (world, 4)
        rank_step_ms = (every[:, 0] / args.steps * 1e3).tolist()
        rank_compute_ms = (every[:, 2] / args.steps * 1e3).tolist()
        elapsed = every[:, 0].max().item()
        if sustained:
            sustained = (every[:, 1].max().item(), sustained[1])
        compute_ms = every[:, 2].max().item() / args.steps * 1e3
        gather_ms = every[:, 3].max().item() / args.steps * 1e3

    if rank == 0:
        total_samples = world * samples_per_step * args.steps
        value = total_samples / elapsed
        per_gpu = value / world
        dtype = 'fp32' if fargan else args.dtype   # the arithmetic type
        if fargan:
            workload = (
                f'Generator.forward (prepare_features + FARGAN, '
                f'config/fargan.py), batch {args.batch} x {args.seconds:g} s '
                f'per GPU ({frames} frames = {frames * 4} dependent sub-frame '
                f'steps), random-init weights stored as '
                f'{promonet_amd.FARGAN_WEIGHT_DTYPE}, fp32 arithmetic')
        elif stand_in:
            workload = (
                f'STAND-IN torch function (not the HIP engine): plumbing test '
                f'of the {world}-rank harness, batch {args.batch} x '
                f'{args.seconds:g} s per rank - NOT a measurement')
        else:
            workload = (
                f'Generator.forward (prepare_features + HiFi-GAN), '
                f'batch {args.batch} x {args.seconds:g} s per GPU '
                f'({frames} frames, {frames * 256} samples each), '
                f'random-init weights, {args.dtype} MFMA operands, '
                f'fp32 accumulate and activations')
        result = {
            'metric': f'audio samples/sec (22.05 kHz), batch-{args.batch} '
                      f'{args.seconds:g} s utterances',
            'value': value,
            'unit': 'samples/s',
            'n_gpus': world,
            'world_size': dist.get_world_size() if world > 1 else 1,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3,
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': dtype,
            'accuracy': (
                'fp32 arithmetic; weights stored as '
                f'{promonet_amd.FARGAN_WEIGHT_DTYPE} (max-abs against the '
                'reference over a whole 10 s utterance, random-init weights: '
                'fp32 2e-6, mixed 6e-6, f16 6.7e-5 - '
                'tests/test_gpu_fargan.py::test_full_size_config5)'
                if fargan else
                f'{args.dtype} operands at the 1e-4 max-abs gate on RANDOM-INIT '
                'weights (audio peak 0.017: tests/test_gpu_model.py::'
                'test_full_size_every_sample); at trained-checkpoint scale see '
                'DESIGN.md section 3 - the library default is the split-f16 '
                "'checkpoint' mode (--dtype checkpoint)"),
            'data': 'synthetic' if not stand_in else
                    'synthetic; stand-in model (plumbing test, invalid as a '
                    'measurement)',
            'config': {
                'workload': workload,
                'model': args.model,
                'batch_per_gpu': args.batch,
                'frames': frames,
                'backend': backend,
                'devices': torch.cuda.device_count()
                           if torch.cuda.is_available() else 0,
                'parallelism': f'batch-sharded x{world}' + (
                    ' + RCCL all-gather of audio (overlapped with the next '
                    'step, drained inside the timed region)' if overlap else
                    ' + all-gather of audio staged through host memory over '
                    'gloo (ranks share a GPU: test-box mode, NOT a scaling '
                    'measurement)' if gather and device.type == 'cuda' else
                    ' + all-gather of audio over gloo (host tensors)'
                    if gather else '')},
            'rtf': value / promonet_amd.SAMPLE_RATE,
            'samples_per_sec_per_gpu': per_gpu,
            'rtf_per_gpu': per_gpu / promonet_amd.SAMPLE_RATE,
        }
        if world > 1:
            # (the timed region's max-over-ranks step against the same steps
            # without the collective: what of the gather is NOT hidden)
            rccl = None
            try:
                rccl = '.'.join(str(v) for v in torch.cuda.nccl.version())
            except Exception:
                pass
            result['multi_gpu'] = {
                'world_size_reported_by_backend': dist.get_world_size(),
                # data plane (weight broadcast, audio all-gather): 'nccl' =
                # RCCL when every rank owns a GPU; the control plane
                # (barriers, this timing reduction) is always gloo
                'backend': backend,
                'control_plane_backend': dist.get_backend(),
                'ranks_share_a_device': promonet_amd.distributed.folded(),
                'gloo_fallback': promonet_amd.distributed._STATE['note'],
                # what every rank bound: device index, PCI / UUID identity
                'devices': promonet_amd.distributed.rank_devices(),
                'rccl_version': rccl,
                'compute_ms_per_step': compute_ms,
                'gather_alone_ms_per_step': gather_ms,
                'exposed_gather_ms': elapsed / args.steps * 1e3 - compute_ms,
                'rank_ms_per_step': rank_step_ms,
                'rank_ms_per_step_min': min(rank_step_ms),
                'rank_ms_per_step_max': max(rank_step_ms),
                'rank_compute_ms_per_step': rank_compute_ms,
                'gathered_bytes_per_rank_per_step':
                    world * samples_per_step * 4 if gather else 0,
                'hang_watchdog_seconds': watchdog.seconds}
        if sustained:
            seconds, count = sustained
            result['sustained_ms_per_step'] = seconds / count * 1e3
            result['sustained_steps'] = count
            result['sustained_value'] = \
                world * samples_per_step * count / seconds
        if fargan:
            times = [a.elapsed_time(b) for a, b in forward_events]
            result['roofline'] = fargan_roofline(
                promonet_amd.FARGAN_WEIGHT_DTYPE, args.batch, frames,
                sum(times) / len(times), per_gpu)
            if not args.no_traffic and world == 1:
                # HBM bytes the cluster kernel really moved (rocprofv3 PMC,
                # two passes over one forward of this command)
                table, source = measure_traffic(args, timeout=300.)
                if table and 'fargan_cluster' in table:
                    result['roofline']['traffic'] = table['fargan_cluster']
                    result['roofline']['traffic_source'] = source
                    result['roofline']['traffic_all_kernels'] = table
                    result['roofline']['traffic_note'] = (
                        'memory-side requests of the 6 agent-scope exchanges '
                        'per step (write-through granule stores and polling '
                        'loads that bypass the L2 by construction), ~24 MB a '
                        'step: the same for fp32 and mixed weight storage '
                        '(profiles/r06/fargan), i.e. not weight re-reads - '
                        'those stay under the L2 (an eighth of the weights '
                        'per XCD)')
                else:
                    result['roofline']['traffic_source'] = \
                        f'not measured: {source}'

        elif stand_in:
            result['stand_in'] = True
            result['roofline'] = None
        else:
            # dominant kernel family: HIP events around its launches, on the
            # launch stream, inside the timed region; every other kernel's row
            # comes from the separate per-kernel pass (`profile`, per
            # `table_steps` steps - scaled to args.steps below)
            label, row = max(timed_profile.items(), key=lambda kv: kv[1]['ms'])
            scale = args.steps / table_steps
            profile = {
                k: {'launches': int(round(v['launches'] * scale)),
                    'ms': v['ms'] * scale, 'flops': v['flops'] * scale,
                    'bytes': v['bytes'] * scale}
                for k, v in kernel_table.items()}
            profile[label] = row
            avg_ms = row['ms'] / row['launches']
            flops_per_launch = row['flops'] / row['launches']
            achieved = flops_per_launch / (avg_ms * 1e-3) / 1e12
            traffic, measured_step_bytes, covered_ms = None, None, 0.
            operand = operand_type_of(label, args.dtype)
            sustained_peak = None
            if world == 1:
                try:
                    sustained_peak = mfma_probe(operand, device)
                    if sustained_peak and operand in ('f16x3', 'f16a2'):
                        # per USEFUL flop (three / two MFMAs per product)
                        sustained_peak /= 3. if operand == 'f16x3' else 2.
                except Exception as error:          # noqa: BLE001
                    sys.stderr.write(f'bench.py: mfma probe failed: {error}\n')
            table, traffic_source = None, None
            if not args.no_traffic and world == 1:
                table, traffic_source = measure_traffic(args)
                if table is not None:
                    table = {f'{k}:{args.dtype}': v for k, v in table.items()}
            traffic_file = ROOT / 'profiles' / 'traffic.json'
            if table is None and traffic_file.exists() and \
                    args.batch == 32 and frames == 861:
                table = json.loads(traffic_file.read_text())
                traffic_source = (
                    'profiles/traffic.json: rocprofv3 PMC of this build '
                    '(FETCH_SIZE x 2 + WRITE_SIZE, separate passes), not '
                    f're-measured in this run ({traffic_source or "--no-traffic"})')
            if table is not None:
                traffic = table.get(f'{label}:{args.dtype}')
                measured_step_bytes = 0.
                for key, entry in profile.items():
                    nbytes = table.get(f'{key}:{args.dtype}')
                    if nbytes is None:
                        # no PMC pass for this kernel: its algorithmic bytes
                        nbytes = entry['bytes'] / entry['launches']
                    else:
                        covered_ms += entry['ms']
                    measured_step_bytes += \
                        nbytes * entry['launches'] / args.steps
            kernel_ms = sum(r['ms'] for r in profile.values()) / args.steps
            result['roofline'] = {
                'kernel': label,
                'bound': 'mfma',
                'achieved': achieved,
                'operands': operand,
                'peak': PEAK_TFLOPS[operand],
                'unit': 'TFLOP/s',
                'frac': achieved / PEAK_TFLOPS[operand],
                'traffic': traffic,
                'traffic_source': traffic_source if traffic else None,
                # the matrix pipe's rate on a register-resident loop of the same
                # MFMA instruction, measured in THIS run right after the timed
                # steps (pm_mfma_probe, ~50 ms): what the power cap leaves
                'sustained_peak': sustained_peak,
                'sustained_peak_source': 'pm_mfma_probe in this run'
                                         if sustained_peak else None,
                'frac_of_sustained_peak': (
                    achieved / sustained_peak if sustained_peak else None),
                'avg_launch_ms': avg_ms,
                'launches_per_step': row['launches'] // args.steps,
                'algorithmic_flops_per_launch': flops_per_launch,
                'algorithmic_bytes_per_launch': row['bytes'] / row['launches'],
                'algorithmic_gbs': row['bytes'] / row['launches'] /
                                   (avg_ms * 1e-3) / 1e9,
                'share_of_kernel_time': row['ms'] / sum(
                    r['ms'] for r in profile.values())}
            if world > torch.cuda.device_count():
                # (test-box mode: the HIP events of rank 0 also span the other
                # ranks' kernels on the shared GPU)
                result['roofline']['note'] = (
                    f'{world} ranks folded onto {torch.cuda.device_count()} '
                    'GPU(s): per-kernel timings include the other ranks\' '
                    'launches - plumbing check, not a roofline measurement')
            result['whole_path'] = {
                'tflops': per_gpu * FLOP_PER_SAMPLE / 1e12,
                'frac_of_mfma_peak': per_gpu * FLOP_PER_SAMPLE / 1e12 /
                                     PEAK_TFLOPS[operand_type_of('input_conv', args.dtype)],
                # HBM bytes the kernels really moved per step (rocprofv3 PMC,
                # FETCH_SIZE x 2 + WRITE_SIZE, profiles/traffic.json) over
                # this run's step time: the ACHIEVED HBM rate
                'measured_hbm_gbs': (
                    measured_step_bytes / (elapsed / args.steps) / 1e9
                    if measured_step_bytes else None),
                'measured_frac_of_hbm_peak': (
                    measured_step_bytes / (elapsed / args.steps) / 1e9 /
                    PEAK_HBM_GBS if measured_step_bytes else None),
                'pmc_covered_share_of_kernel_time': (
                    covered_ms / sum(r['ms'] for r in profile.values())
                    if measured_step_bytes else None),
                # a MODEL, not a measurement: the bytes an unfused
                # layer-by-layer implementation would move
                # at this step time; the fused kernels move far fewer
                'unfused_layer_model_gbs':
                    per_gpu * ELEMENTS_PER_SAMPLE * 4 / 1e9,
                'kernel_ms_per_step': kernel_ms}
            result['kernels_source'] = (
                f'{label}: HIP events inside the timed region; the other rows: a '
                f'separate pass of {table_steps} steps with events around every '
                'launch (outside `value`)')
            result['kernels'] = {
                k: {'ms_per_step': v['ms'] / args.steps,
                    'launches_per_step': v['launches'] // args.steps,
                    'tflops': v['flops'] / max(v['ms'], 1e-9) / 1e9,
                    'gbs': v['bytes'] / max(v['ms'], 1e-9) / 1e6}
                for k, v in sorted(profile.items())}
        # (the secondary block rides on the HiFi-GAN headline line only: it
        # builds its own models under the default configuration)
        if world == 1 and hifigan and not args.no_secondary:
            # (the headline model's workspace goes back to the allocator first)
            del model, inputs
            torch.cuda.empty_cache()
            result['secondary'] = secondary_block(device)
        if world == 1 and not stand_in and not args.no_cpu_baseline:
            result['cpu_baseline'] = cpu_baseline(args.model)
        print(json.dumps(result))

    if world > 1:
        watchdog.arm('final barrier')
        dist.barrier()
        promonet_amd.distributed.shutdown()
    watchdog.disarm()


if __name__ == '__main__':
    main()
