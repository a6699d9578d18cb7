"""Synthetic inputs of BASELINE.md section 2, generated directly in device memory.

SplitMix64 -> U[0,1): u_i = (mix(seed + (i+1)*GOLDEN) >> 11) * 2^-53, the same stream as
oracle.splitmix_uniform (checked in tests/test_synth_cpu.py), written with wrapping int64 torch ops so that a
1e8-element query vector never crosses PCIe.
"""
import math

GOLDEN = 0x9E3779B97F4A7C15
C1 = 0xBF58476D1CE4E5B9
C2 = 0x94D049BB133111EB


def _s64(v):
    v &= (1 << 64) - 1
    return v - (1 << 64) if v >= (1 << 63) else v


def _lsr(z, k):
    """logical right shift of an int64 tensor"""
    return (z >> k) & ((1 << (64 - k)) - 1)


def splitmix_uniform(seed, n, device, chunk=1 << 24, offset=0):
    """elements offset .. offset+n-1 of the stream (offset > 0: a rank's contiguous shard of one global query set)"""
    import torch
    out = torch.empty(n, dtype=torch.float64, device=device)
    for a in range(0, n, chunk):
        b = min(n, a + chunk)
        i = torch.arange(offset + a + 1, offset + b + 1, dtype=torch.int64, device=device)
        z = i * _s64(GOLDEN) + _s64(seed)
        z = (z ^ _lsr(z, 30)) * _s64(C1)
        z = (z ^ _lsr(z, 27)) * _s64(C2)
        z = z ^ _lsr(z, 31)
        out[a:b] = _lsr(z, 11).to(torch.float64) * (2.0 ** -53)
    return out


def config_grid(ng):
    """X_i = i/(ng-1), Y_i = sin(2 pi X_i) + 0.5 X_i  (BASELINE configs 1-2), as numpy float64."""
    import numpy as np
    X = np.arange(ng) / (ng - 1)
    return X, np.sin(2 * math.pi * X) + 0.5 * X


def config3_table(n, device):
    """Z(i,j) = sin(2 pi y_i) cos(2 pi x_j) + x_j y_i on uniform [0,1]^2, column-major flat (ny*nx) on device."""
    import torch
    ax = torch.arange(n, dtype=torch.float64, device=device) / (n - 1)
    z = torch.sin(2 * math.pi * ax)[:, None] * torch.cos(2 * math.pi * ax)[None, :] + ax[None, :] * ax[:, None]
    return z.t().contiguous().reshape(-1)      # [j*ny + i]
