import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_group(name):
    """Read one tests/golden/*.npz trajectory group -> {trajectory name: {field: array}}."""
    z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    out = {}
    for n in z["names"]:
        n = str(n)
        out[n] = {k.split("/", 1)[1]: z[k] for k in z.files if k.startswith(n + "/")}
    return out


def all_trajectories():
    items = []
    for f in ("f1_golden_episode.npz", "f2_seeded.npz", "f3_branches.npz"):
        for n, t in load_group(f).items():
            items.append(pytest.param(t, id=f"{f[:2]}-{n}"))
    return items


def needs_raw_state(t):
    """F1 (recorded start has psi1 != psi2) and the artificial F3 case put a raw state back."""
    return "recorded_states" in t or bool(t.get("raw_state_override", False))


@pytest.fixture(scope="session")
def gpu_device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
