import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = np.load(os.path.join(GOLDEN_DIR, name + '.npz'))
        return cache[name]
    return load


# (V, settings name, weight seed) and input recipes of every golden case: the single
# source of truth shared by tests/golden/make_golden.py and the parity tests.
CASES = {
    'tiny': dict(V=64, settings='tiny', wseed=1, B=6, R=6, T=8, in_seed=11, s2s_B=4, s2s_seed=12),
    'cfg1': dict(V=10000, settings='default', wseed=0, B=4, R=36, T=20, in_seed=1, s2s_B=5, s2s_seed=2),
    'b128': dict(V=10000, settings='default', wseed=0, B=128, R=36, T=20, in_seed=21, s2s_B=80, s2s_seed=22),
}


def case_setup(name):
    from insenticap_model_amd import synth
    c = CASES[name]
    st = synth.TINY_SETTINGS if c['settings'] == 'tiny' else synth.DEFAULT_SETTINGS
    w = synth.make_weights(c['V'], st, seed=c['wseed'])
    d = synth.make_inputs(c['B'], c['V'], st, regions=c['R'], seq_len=c['T'], seed=c['in_seed'])
    s2s = synth.make_inputs(c['s2s_B'], c['V'], st, regions=c['R'], seq_len=c['T'], seed=c['s2s_seed'])
    return c, st, w, d, s2s


def trusted_prefix(margins, masks, thresh):
    """Per row: number of leading steps whose golden top-1/top-2 margin exceeds `thresh`
    (assert token equality only where the margin is far above fp32 noise;
    once a near-tie flips, the rest of that row legitimately diverges)."""
    B, T = margins.shape
    n = np.zeros(B, dtype=np.int64)
    for b in range(B):
        t = 0
        while t < T and (masks[b, t] == 0 or margins[b, t] > thresh):
            t += 1
        n[b] = t
    return n


def digest(a):
    """Same fingerprint as make_golden.grad_digest: sum, abs-sum, l2, 64 strided samples."""
    flat = np.asarray(a, dtype=np.float64).reshape(-1)
    idx = (np.arange(64, dtype=np.int64) * 2654435761 % flat.size)
    return np.concatenate([[flat.sum(), np.abs(flat).sum(), np.sqrt((flat ** 2).sum())], flat[idx]])


def assert_digest_close(got, ref, name, rel=1e-4, floor=1e-7):
    """`floor` covers tensors whose true value is 0 (e.g. the gradient of a bias that a
    softmax cancels), where only rounding noise of order 1e-9 is left."""
    n_floor = floor * 64
    assert abs(got[2] - ref[2]) <= rel * ref[2] + n_floor, (name, 'l2', got[2], ref[2])
    assert abs(got[0] - ref[0]) <= rel * ref[1] + n_floor, (name, 'sum', got[0], ref[0])
    np.testing.assert_allclose(got[3:], ref[3:], atol=rel * np.abs(ref[3:]).max() + floor, err_msg=name)
