"""A short, seeded run of tools/fuzz_gemm_paths.py: random shapes through the forward GEMM entry points on both engines."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tools'))


@pytest.mark.parametrize('seed', [3, 11])
def test_random_shapes_against_fp64(seed):
    import fuzz_gemm_paths
    assert fuzz_gemm_paths.run(seed, 30, verbose=False) == 0
