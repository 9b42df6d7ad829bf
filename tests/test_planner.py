"""Radix planning: the product's host logic (libmifft, no GPU needed) against the oracle's
restatement and against values worked out from the reference rules
(fft/fft/_utils.mojo:125-221, fft/fft/fft.mojo:49-104)."""
import pytest

import hackathon_fft_amd as mf
from conftest import load_matrix
from oracle import mifft_oracle as O

EXPECTED_ORDERED = {
    (1024, (2,)): [2] * 10,
    (128, (2,)): [2] * 7,
    (93, (3, 31)): [31, 3],
    (48, (3, 2)): [3, 2, 2, 2, 2],
    (20, (5, 2)): [5, 2, 2],
    (100, (10,)): [10, 10],
    (16, (2, 4)): [4, 4],
    (60, (3, 4, 5)): [5, 4, 3],
    (6, (2, 3)): [3, 2],
    (128, (16, 8)): [16, 8],
    (64, (4,)): [4, 4, 4],
}


@pytest.mark.parametrize("key", list(EXPECTED_ORDERED))
def test_ordered_bases_known(key):
    n, bases = key
    assert mf.ordered_bases(n, bases) == EXPECTED_ORDERED[key]
    assert O.ordered_bases(n, bases) == EXPECTED_ORDERED[key]


@pytest.mark.parametrize("n,bases", load_matrix())
def test_ordered_bases_reference_matrix(n, bases):
    got = mf.ordered_bases(n, bases)
    assert got == O.ordered_bases(n, bases)
    prod = 1
    for r in got:
        prod *= r
    assert prod == n and got == sorted(got, reverse=True)


@pytest.mark.parametrize("n", list(range(2, 200)) + [256, 480, 640, 729, 1000, 1024, 2048, 4096, 16384, 30030])
@pytest.mark.parametrize("target", ["cpu", "gpu"])
def test_estimate_bases_matches_oracle(n, target):
    assert mf.estimate_best_bases(n, target) == O.estimate_bases(n, target)


def test_estimate_bases_baseline_configs():
    assert mf.ordered_bases(1024, mf.estimate_best_bases(1024, "gpu")) == [2] * 10
    assert mf.ordered_bases(128, mf.estimate_best_bases(128, "gpu")) == [2] * 7
    assert mf.ordered_bases(93, mf.estimate_best_bases(93, "gpu")) == [31, 3]
    assert mf.ordered_bases(480, mf.estimate_best_bases(480, "gpu")) == [5, 3, 2, 2, 2, 2, 2]
    assert mf.ordered_bases(640, mf.estimate_best_bases(640, "gpu")) == [5, 2, 2, 2, 2, 2, 2, 2]
    assert mf.estimate_best_bases(93, "cpu") == [3, 31]


@pytest.mark.parametrize("n,bases,status", [
    (32, (4, 2), -5),     # greedy over-shoots: 4,4 then five 2s (reference compile-time assert)
    (12, (5,), -5),       # base does not divide
    (8, (1, 8), -6),      # base 1
    (8, (), -7),          # empty
    (202, (2,), -5),      # incomplete
])
def test_bad_bases(n, bases, status):
    with pytest.raises(mf.MifftError) as ei:
        mf.ordered_bases(n, bases)
    assert ei.value.status == status
    with pytest.raises(O.OracleError) as eo:
        O.ordered_bases(n, bases)
    assert eo.value.status == status


def test_prime_factor_above_97_is_rejected_like_the_reference():
    # _estimate_best_bases returns an incomplete list (fft/fft/fft.mojo:88-104) -> later assert
    bases = mf.estimate_best_bases(2 * 101, "cpu")
    with pytest.raises(mf.MifftError):
        mf.ordered_bases(2 * 101, bases)
