"""CLI of promonet/synthesize/__main__.py (same flags)."""
import argparse
from pathlib import Path

import promonet_amd


def parse_args():
    parser = argparse.ArgumentParser(description='Synthesize speech')
    parser.add_argument('--loudness_files', type=Path, nargs='+', required=True)
    parser.add_argument('--pitch_files', type=Path, nargs='+', required=True)
    parser.add_argument(
        '--periodicity_files', type=Path, nargs='+', required=True)
    parser.add_argument('--ppg_files', type=Path, nargs='+', required=True)
    parser.add_argument('--output_files', type=Path, nargs='+', required=True)
    parser.add_argument('--speakers', type=int, nargs='+')
    parser.add_argument('--spectral_balance_ratio', type=float, default=1.)
    parser.add_argument('--loudness_ratio', type=float, default=1.)
    parser.add_argument('--checkpoint', type=Path)
    parser.add_argument('--gpu', type=int, required=True)
    return parser.parse_args()


if __name__ == '__main__':
    promonet_amd.synthesize.from_files_to_files(**vars(parse_args()))
