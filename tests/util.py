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
