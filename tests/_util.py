"""Shared helpers for the tests: golden loading, state templates, digests."""
import ast
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_DT = {"torch.float32": torch.float32, "torch.int64": torch.int64}


def golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def state_template(fusion: str):
    """Zero state with the reference's keys / shapes / dtypes (from pins.npz)."""
    pins = golden("pins.npz")
    st = {}
    for k, s, d in zip(pins[f"{fusion}_keys"], pins[f"{fusion}_shapes"], pins[f"{fusion}_dtypes"]):
        st[str(k)] = torch.zeros(ast.literal_eval(str(s)), dtype=_DT[str(d)])
    for k in st:
        if k.endswith(("x_range", "y_range")):
            st[k] = torch.tensor([-50, 50])
    return st


def digest(t: torch.Tensor) -> np.ndarray:
    t = t.detach().double().cpu().reshape(-1)
    head = torch.zeros(4, dtype=torch.float64)
    head[: min(4, t.numel())] = t[:4]
    return np.concatenate([[t.sum().item(), t.norm().item(), t.abs().max().item()], head.numpy()])


def digest_close(got: np.ndarray, want: np.ndarray, rtol=2e-4, atol=2e-6):
    """Compare two digests: the sum against the tensor's norm scale, the rest relatively."""
    scale = max(abs(want[1]), 1e-12)
    ok = abs(got[0] - want[0]) <= rtol * scale + atol
    ok &= abs(got[1] - want[1]) <= rtol * scale + atol
    ok &= abs(got[2] - want[2]) <= rtol * max(abs(want[2]), 1e-12) + atol
    ok &= bool(np.all(np.abs(got[3:] - want[3:]) <= rtol * max(abs(want[2]), 1e-12) + atol))
    return bool(ok)
