"""Device index stream (csrc/rng.hip) == libstdc++ mt19937 + uniform_int_distribution<long>,
bit for bit, through the C ABI (cymf_rng_fill_uniform)."""
import numpy as np
import pytest

import oracle
from conftest import golden
from cymf_amd import _lib

pytestmark = pytest.mark.gpu


def test_stream_vs_libstdcxx_fixture():
    g = golden("index_stream")
    checked = 0
    for key in g.files:
        seed, rng_range = (99, 100000) if key.startswith("seed99") else (1234, int(key[1:]))
        want = g[key]     # ranges 2**32, 2**32 + 12345, 5 * 2**32 + 3 included: the non-Lemire branches of libstdc++
        got = _lib.rng_fill_uniform(seed, rng_range, len(want))
        assert np.array_equal(got, want), key
        checked += 1
    assert checked >= 11


def test_stream_survey_known_answers():
    assert _lib.rng_fill_uniform(1234, 3706, 12).tolist() == [709, 1844, 2305, 3030, 1622, 2268, 2910, 2858, 2890, 3189, 1010, 558]


@pytest.mark.parametrize("rng_range,n,skip", [(100000, 300000, 0), (100000, 5000, 1234567), (1682, 70000, 623),
                                              (3000000000, 200000, 17), (1, 1000, 0), (2**32 - 1, 5000, 3)])
def test_stream_vs_oracle(rng_range, n, skip):
    # ranges near 2^32 reject ~30% of the raw words: the ordered-compaction path of the kernel
    got = _lib.rng_fill_uniform(1234, rng_range, n, skip=skip)
    assert np.array_equal(got, oracle.uniform_stream(1234, rng_range, n, skip=skip))


def test_stream_block_boundaries():
    # requests that end exactly at / one off a 624-word block boundary, then continue
    full = oracle.uniform_stream(7, 1000, 624 * 5 + 10)
    for n in (623, 624, 625, 1248, 1249):
        assert np.array_equal(_lib.rng_fill_uniform(7, 1000, n), full[:n])
        assert np.array_equal(_lib.rng_fill_uniform(7, 1000, 7, skip=n), full[n:n + 7])


@pytest.mark.parametrize("rng_range,n,skip", [(2**32, 3000, 0), (2**32, 700, 1300), (2**32 + 1, 5000, 11), (3 * 2**32 - 1, 4000, 623),
                                              (2**33, 2500, 1), (20000 * 300000, 3000, 0), (2**40 + 12345, 2000, 77)])
def test_ranges_of_2_to_32_and_more_vs_oracle(rng_range, n, skip):
    """RelMF draws cells from UniformGenerator(0, U*I) on long (cymf/relmf.pyx:128): at 2^32 libstdc++ takes one raw word,
    above it a Lemire-drawn high part times 2^32 plus a raw word, the pair redrawn while it exceeds the range
    (bits/uniform_int_dist.h:325-347; 2^32 + 1 rejects half of the pairs).  skip lands inside blocks and pairs."""
    got = _lib.rng_fill_uniform(1234, rng_range, n, skip=skip)
    assert np.array_equal(got, oracle.uniform_stream(1234, rng_range, n, skip=skip))
    assert got.max() < rng_range and got.min() >= 0


def test_range_limits_are_errors():
    with pytest.raises(_lib.CymfError):
        _lib.rng_fill_uniform(1234, 2**62 + 1, 10)
    with pytest.raises(_lib.CymfError):
        _lib.rng_fill_uniform(1234, 0, 10)
    assert len(_lib.rng_fill_uniform(1234, 10, 0)) == 0


@pytest.mark.parametrize("rng_range,n,skip", [(100000, 9_000_000, 0), (100000, 200_000, 4_193_280 * 2 - 100_000),
                                              (3706, 5_000_000, 4_193_279), (40_000_000, 6_000_000, 11),
                                              (160_000_000, 3_000_000, 6_000_000), (300_000_000, 4_500_000, 0)])
def test_chunked_jump_ahead_generator_vs_oracle(rng_range, n, skip):
    """skip + n >= 4M selects the parallel generator: chunk start states by the MT19937 jump-ahead
    polynomials (tools/gen_mt_jump.py), one workgroup per 1,048,320-word chunk (4,193,280 = 4 chunks), ordered gather.
    range 4e7 rejects ~1% of the raw words; 1.6e8 (the U*I of a 20000 x 8000 RelMF problem) 3.1% and 3e8 2.2%:
    ~100 k rejections per chunk, the rejection list is sized from the range."""
    got = _lib.rng_fill_uniform(1234, rng_range, n, skip=skip)
    want = oracle.uniform_stream(1234, rng_range, n, skip=skip)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("rng_range,n,skip", [(100000, 300_000, 4_193_280 * 16 - 150_000), (100000, 2_000_000, 4_193_280 * 33 + 5),
                                              (3706, 1_000_000, 4_193_280 * 47 - 500_000)])
def test_wide_jump_states_vs_oracle(rng_range, n, skip):
    """Chunk start states come from jumps by 1, 4, 16 and 64 chunks (tools/gen_mt_jump.py: MT_JUMP_POLYS), a run of states per
    launch: draws across the boundary where the 64-chunk jumps take over (chunk 64 = 16 x 4,193,280 words), in the second
    64-run and in the third are the oracle's."""
    got = _lib.rng_fill_uniform(1234, rng_range, n, skip=skip)
    want = oracle.uniform_stream(1234, rng_range, n, skip=skip)
    assert np.array_equal(got, want)
