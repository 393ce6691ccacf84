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


@pytest.fixture(scope="session")
def weights():
    """Synthetic state dicts (chattts_amd.weights recipe); fingerprint-checked against the goldens."""
    from chattts_amd import weights as W

    sds = W.synthetic_all()
    want = {}
    with open(os.path.join(GOLDEN, "weights_fingerprint.txt")) as f:
        for line in f:
            k, v = line.split()
            want[k] = v
    for k in ("embed", "decoder", "vocos", "gpt"):
        got = W.fingerprint(sds[k])
        assert got == want[k], f"synthetic weight recipe '{k}' is not reproducible on this machine/torch build"
    return sds


@pytest.fixture(scope="session")
def golden():
    return {n: np.load(os.path.join(GOLDEN, n + ".npz")) for n in ("sampling", "generate", "generate_big", "generate_params", "generate_stream", "generate_regen", "generate_sweep", "codec", "codec_big", "text")}
