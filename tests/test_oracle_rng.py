"""Pins the oracle of the engine's counter-based generator: Philox4x32-10 known-answer vectors (Random123 kat_vectors,
Salmon et al. SC'11) and the moments of the Box-Muller normals."""
import importlib.util
import os

import numpy as np

from oracle import rng as R

KAT = [
    ((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


def _host_rng():
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("cmbl_host_rng", os.path.join(here, "cmblensing.jl_amd", "rng.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_philox_known_answers():
    host = _host_rng()
    for ctr, key, want in KAT:
        got = R.philox4x32_10(np.array(ctr, dtype=np.uint32), key)
        assert tuple(int(x) for x in got) == want
        assert tuple(host.philox4x32_10(ctr, key)) == want           # the product's scalar host implementation


def test_counter_layout_and_streams():
    a = R.randn(11, 5, 1000)
    assert np.array_equal(a[:37], R.randn(11, 5, 37))               # element j depends on (seed, stream, j) only
    assert not np.allclose(a, R.randn(11, 6, 1000)) and not np.allclose(a, R.randn(12, 5, 1000))
    w = R.philox_words(0x0123456789abcdef, 0xfedcba9876543210, 3)
    k = (0x89abcdef, 0x01234567)
    assert np.array_equal(w[2], R.philox4x32_10(np.array([2, 0, 0x76543210, 0xfedcba98], dtype=np.uint32), k))
    host = _host_rng()
    assert np.allclose(host.uniform(987654321987, 33, 9), R.uniform(987654321987, 33, 9), rtol=0, atol=0)


def test_normal_moments():
    z = R.randn(3, 0, 1 << 20)
    n = z.size
    assert abs(z.mean()) < 4 / np.sqrt(n) and abs(z.var() - 1) < 4 * np.sqrt(2 / n)
    assert abs((z ** 3).mean()) < 4 * np.sqrt(15 / n) and abs((z ** 4).mean() - 3) < 4 * np.sqrt(96 / n)
    assert abs(np.corrcoef(z[:-1], z[1:])[0, 1]) < 4 / np.sqrt(n)
    assert R.randn(3, 0, 64, np.float32).dtype == np.float32
