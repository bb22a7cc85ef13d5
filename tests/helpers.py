"""Shared helpers for the parity tests."""
import numpy as np
import torch


def t(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def cloud_from_golden(g, prefix, nb, feats=True):
    from oracle.cloud import Cloud

    return Cloud([t(g[f"{prefix}_points_{b}"]) for b in range(nb)],
                 [t(g[f"{prefix}_normals_{b}"]) for b in range(nb)],
                 [t(g[f"{prefix}_colors_{b}"]) for b in range(nb)],
                 [t(g[f"{prefix}_feats_{b}"]) for b in range(nb)] if feats else None)


def rel_err(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
