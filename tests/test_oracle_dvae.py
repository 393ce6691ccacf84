"""CPU: the numpy restatement of the full DVAE (oracle/dvae_np.py) against tests/golden/dvae.npz, whose conv trunks were
evaluated by the reference's own `DVAEDecoder` class and whose framing by torch.stft (oracle/make_dvae_goldens.py)."""
import os

import numpy as np
import pytest

from chattts_amd import weights as W
from oracle import dvae_np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["half_second", "odd_frames", "short"]


@pytest.fixture(scope="module")
def nsd():
    sd = W.synthetic_dvae()
    with open(os.path.join(GOLD, "weights_fingerprint.txt")) as f:
        want = dict(line.split() for line in f if line.strip())
    assert W.fingerprint(sd) == want["dvae"]
    return {k: v.numpy() for k, v in sd.items()}


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLD, "dvae.npz"))


@pytest.mark.parametrize("name", CASES)
def test_encode_chain_matches_reference_classes(nsd, gold, name):
    wav = gold[name + ".wav"]
    mel = dvae_np.mel_features(wav, nsd["preprocessor_mel.mel_spec.spectrogram.window"], nsd["preprocessor_mel.mel_spec.mel_scale.fb"])
    assert mel.shape == gold[name + ".logmel"].shape == (1 + wav.size // 256, 100)
    assert np.abs(mel - gold[name + ".logmel"]).max() < 1e-4            # torch.stft framing + filterbank + log
    feat = dvae_np.encoder_features(nsd, gold[name + ".logmel"])
    assert np.abs(feat - gold[name + ".feat"]).max() < 2e-5               # downsample convs + the reference's DVAEDecoder (encoder)
    codes = dvae_np.gfsq_encode(nsd, gold[name + ".feat"])
    assert np.array_equal(codes, gold[name + ".codes"])
    assert np.array_equal(dvae_np.dvae_encode(nsd, wav), gold[name + ".codes"].T)   # the whole chain, [4, T]


@pytest.mark.parametrize("name", CASES)
def test_decode_codes_matches_reference_classes(nsd, gold, name):
    mel = dvae_np.dvae_decode_codes(nsd, gold[name + ".codes"][None].astype(np.int64))[0]
    assert np.abs(mel - gold[name + ".mel_out"]).max() < 2e-5            # GFSQ embed + the reference's DVAEDecoder (decoder) + out_conv


def test_fsq_round_trip_and_ranges(nsd):
    lv = np.array([5, 5, 5, 5])
    rs = np.random.RandomState(0)
    z = rs.standard_normal((1000, 4)).astype(np.float32) * 2
    codes, idx = dvae_np.fsq_quantize(z, lv)
    assert idx.min() >= 0 and idx.max() < 625 and set(np.unique(codes * 2)) <= {-2.0, -1.0, 0.0, 1.0, 2.0}
    assert np.array_equal(dvae_np.fsq_codes_from_index(idx, lv), codes)  # index <-> code bijection
    # embed(encode(x)) is the quantised reconstruction: re-encoding the pre-projection code vector is idempotent
    idx2 = rs.randint(0, 625, size=(50, 4))
    feat = dvae_np.gfsq_embed(nsd, idx2)
    assert feat.shape == (50, 1024) and np.isfinite(feat).all()
    # stride-2 conv identity: equals a dense k4 conv sampled at even positions
    x = rs.standard_normal((1, 11, 8)).astype(np.float32)
    w = rs.standard_normal((6, 8, 4)).astype(np.float32)
    b = rs.standard_normal(6).astype(np.float32)
    y = dvae_np.conv1d_k4s2_cl(x, w, b)
    xp = np.zeros((1, 13, 8), np.float32)
    xp[:, 1:12] = x
    ref = np.stack([sum(xp[0, 2 * t + j] @ w[:, :, j].T for j in range(4)) + b for t in range((11 - 2) // 2 + 1)])
    assert y.shape == (1, 5, 6) and np.abs(y[0] - ref).max() < 1e-5
