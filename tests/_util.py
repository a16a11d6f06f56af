"""Shared helpers for the test-suite (imports the hyphenated package by name)."""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
PKG = "multimodal-hierarchical-cnn-for-sun-salutation-pose-classification_amd"


def pkg(sub=None):
    return importlib.import_module(PKG if sub is None else f"{PKG}.{sub}")


def rel_err(a, b):
    """max|a-b| / max|b|  -- the parity metric of SURVEY.md 8(d)."""
    a = torch.as_tensor(np.asarray(a)).double()
    b = torch.as_tensor(np.asarray(b)).double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def summary(t, nsample=256):
    f = t.detach().double().flatten().cpu()
    stride = max(1, f.numel() // nsample)
    return {"sum": float(f.sum()), "abssum": float(f.abs().sum()),
            "sample": f[::stride][:nsample].float().numpy()}


def check_summary(t, gold, prefix, tol):
    """Compare tensor `t` (NCHW / reference layout) with a golden summary."""
    s = summary(t)
    assert tuple(t.shape) == tuple(int(x) for x in gold[f"{prefix}/shape"]), prefix
    g = gold[f"{prefix}/sample"]
    scale = max(float(np.abs(g).max()), 1e-30)
    assert float(np.abs(s["sample"] - g).max()) / scale <= tol, (prefix, "sample")
    assert abs(s["abssum"] - float(gold[f"{prefix}/abssum"])) <= tol * float(gold[f"{prefix}/abssum"]) + 1e-12, (prefix, "abssum")
