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


def pytest_collection_modifyitems(config, items):
    try:
        import torch
        have_gpu = torch.cuda.is_available()
    except Exception:
        have_gpu = False
    if have_gpu:
        return
    skip = pytest.mark.skip(reason="no HIP device in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
        return cache[name]

    return load


LONG_CHAINS = {"G19_chain1000_ancestral_64": dict(S=64, steps=None, batch=64), "G20_ddim250_128": dict(S=128, steps=250, batch=64),
               "G21_ddim250_256": dict(S=256, steps=250, batch=16), "G22_chain1000_ancestral_128": dict(S=128, steps=None, batch=64),
               "G21b_ddim250_256": dict(S=256, steps=250, batch=16)}


def regenerate_chain_noise(g):
    """The noise of a G19/G20 chain: the reference drew it from torch's global CPU generator (randn for the start image,
    one randn_like per transition), so the fixture stores the seed and a sha256 instead of 16 MB of Gaussians.  Refuses
    (fails) when this torch build does not reproduce the recorded draws."""
    import hashlib
    import torch
    S = g["img_cond"].shape[-1]
    state = torch.random.get_rng_state()
    torch.manual_seed(int(g["noise_seed"]))
    nz = torch.stack([torch.randn((1, 1, S, S)) for _ in range(int(g["n_draws"]))])
    torch.random.set_rng_state(state)
    sha = hashlib.sha256(nz.numpy().tobytes()).digest()
    assert sha == bytes(bytearray(g["noise_sha256"].tolist())), \
        "torch.randn does not reproduce the fixture's recorded noise on this build: regenerate with tools/make_goldens.py"
    probe = nz[[0, 1, len(nz) - 1]].reshape(3, -1)[:, :8].numpy()
    assert np.array_equal(probe, g["noise_probe"])
    return nz
