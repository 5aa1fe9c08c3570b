"""File-level throughput on the GPU box: N synthetic
utterances on disk -> wav files through `synthesize.from_files_to_files_batched`
(length-sorted ragged batches) and, for a few files, through the reference-style sequential
`from_files_to_files`. usage: python scripts/bench_files.py [files] [seconds]"""
import json
import sys
import tempfile
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / 'oracle'))
import restatement as oracle  # noqa: E402  (test infrastructure: inputs, weights)
import promonet_amd  # noqa: E402


def main():
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 10.
    device = torch.device('cuda:0')
    promonet_amd.configure(COMPUTE_DTYPE='bf16')
    model = promonet_amd.model.Generator()
    model.load_state_dict(oracle.random_state(seed=0))
    model = model.to(device).eval()
    promonet_amd.synthesize.set_model(model, device)

    gen = torch.Generator().manual_seed(0)
    with tempfile.TemporaryDirectory() as tmp:
        tmp = Path(tmp)
        names = {key: [] for key in ('loudness', 'pitch', 'periodicity', 'ppg', 'out', 'seq')}
        total_samples = 0
        for index in range(count):
            # utterances of 60 % ... 100 % of `seconds`
            frames = int(seconds * 22050 / 256 * (.6 + .4 * torch.rand(1, generator=gen).item()))
            loudness, pitch, periodicity, ppg, *_ = oracle.synthetic_inputs(1, frames, seed=index)
            torch.save(loudness[0], tmp / f'{index}-loudness.pt')
            torch.save(pitch, tmp / f'{index}-pitch.pt')
            torch.save(periodicity, tmp / f'{index}-periodicity.pt')
            torch.save(ppg[0], tmp / f'{index}-ppg.pt')
            for key in ('loudness', 'pitch', 'periodicity', 'ppg'):
                names[key].append(tmp / f'{index}-{key}.pt')
            names['out'].append(tmp / 'batched' / f'{index}.wav')
            names['seq'].append(tmp / 'sequential' / f'{index}.wav')
            total_samples += frames * 256
        args = [names[k] for k in ('loudness', 'pitch', 'periodicity', 'ppg')]
        result = {'files': count, 'audio_seconds': total_samples / 22050}
        # warm-up (weights packed, kernels loaded)
        promonet_amd.synthesize.from_files_to_files_batched(
            *[a[:32] for a in args], names['out'][:32], gpu=0, batch_size=32)
        # (best of 3: the first pass also warms the page cache)
        best = {}
        for _ in range(3):
            for label, kwargs in (('batched', {}),
                                  ('batched_in_process', {'num_workers': 0})):
                start = time.perf_counter()
                promonet_amd.synthesize.from_files_to_files_batched(
                    *args, names['out'], gpu=0, batch_size=32, **kwargs)
                torch.cuda.synchronize()
                elapsed = time.perf_counter() - start
                best[label] = min(best.get(label, elapsed), elapsed)
        for label, elapsed in best.items():
            result[label] = {
                'seconds': elapsed, 'files_per_s': count / elapsed,
                'rtf': total_samples / 22050 / elapsed}
        few = min(count, 32)
        few_samples = sum(
            torch.load(names['pitch'][i]).shape[-1] * 256 for i in range(few))
        start = time.perf_counter()
        promonet_amd.synthesize.from_files_to_files(
            *[a[:few] for a in args], names['seq'][:few], gpu=0)
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - start
        result['sequential'] = {
            'files': few, 'seconds': elapsed, 'files_per_s': few / elapsed,
            'rtf': few_samples / 22050 / elapsed}
    print(json.dumps(result))


if __name__ == '__main__':
    # (the batched path spawns worker processes, which re-import this module)
    main()
