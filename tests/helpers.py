"""Shared helpers for the parity tests (test infrastructure)."""
import glob
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    nb = np.linalg.norm(b)
    return float(np.linalg.norm(a - b) / (nb if nb > 0 else 1.0))


def golden_files(prefix):
    return sorted(glob.glob(os.path.join(GOLDEN, prefix + "*.npz")))


def golden_state_dict(g, dtype=torch.float32):
    """state_dict stored in a schnet_ref_*.npz fixture, with the `conv.nn.*` aliases of `mlp.*` re-added."""
    sd = {}
    for k in g.files:
        if k.startswith("sd:"):
            name = k[3:]
            v = torch.from_numpy(g[k])
            sd[name] = v.to(dtype) if v.is_floating_point() else v
            if ".mlp." in name:
                sd[name.replace(".mlp.", ".conv.nn.")] = sd[name]
    return sd
