"""Helpers shared by the parity tests (layout conversion, error metrics)."""
import torch


def pad32(c):
    return (c + 31) // 32 * 32


def to_cl(x, cpad=None):
    """(B, C, L) -> channels-last (B, L, C_pad), zero padded."""
    b, c, l = x.shape
    cpad = cpad or pad32(c)
    out = torch.zeros(b, l, cpad, dtype=x.dtype, device=x.device)
    out[:, :, :c] = x.permute(0, 2, 1)
    return out.contiguous()


def from_cl(x_cl, c):
    return x_cl[:, :, :c].permute(0, 2, 1).contiguous()


def max_abs(a, b):
    return (a.double().cpu() - b.double().cpu()).abs().max().item()


def rel_err(a, b):
    scale = b.double().abs().max().item()
    return max_abs(a, b) / max(scale, 1e-30)


# Measured-error ledger: every parity assertion goes through `check`, which
# keeps the largest error seen per (kind, dtype). With PM_RECORD_ERRORS set the
# ledger is written to gpurun_out/measured_errors.json at session end
# (tests/conftest.py) - the gates in the test files are set from it (<= 3x the
# measured value), so a regression that triples an error fails a unit test.
MEASURED = {}


def check(error, tolerance, kind, detail=None):
    key = kind
    MEASURED[key] = max(MEASURED.get(key, 0.), float(error))
    assert error < tolerance, (kind, error, tolerance, detail)
