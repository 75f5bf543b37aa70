"""Oracle hardening: is the float32 idf of term_weighting.go:37 sensitive to WHICH math.Log ran?

The oracle restates Go's pure-Go math.Log/Log2 (log.go, log10.go).  go1.12 on amd64 dispatches math.Log to
log_amd64.s, a transcription of the same algorithm; neither source is in /root/reference, so the claim
"idf bit-identical to the reference" must not depend on the last bit of the float64 logarithm.  These tests show
that it does not for every document frequency the test and benchmark indexes can hold:

  * float32(orc_go_log2(N/df)) equals the correctly rounded float32 of log2(N/df) (80-bit log2l, spot-checked
    against 60-digit mpmath), and
  * orc_go_log2(N/df) is never within 4 x (ulp(result) + ulp(Log(frac))/ln 2) of a float32 rounding boundary,
    so no libm whose Log differs from log.go's in the last ulps can produce another idf.

Any df where that fails is listed (DESIGN.md §1 quotes the list).
"""
import numpy as np
import pytest

from oracle import pyoracle

# every N (= total_docs) the tests and bench.py pass to the TF-IDF build, with the largest df an index of that
# size holds (synth.zipf_df clips df at N/4; hand-built tables may go up to N, and Q7 — total_docs != #indexed
# docs — up to 2N in the KATs)
CASES = [
    (10_000_000, 2_500_000),      # BASELINE config 3/5 (bench.py, tests/test_gpu_fullsize.py): clip_frac 0.25
    (1_000_000, 1_000_000), (1 << 20, 1 << 20), (200_000, 200_000), (100_000, 100_000), (20_000, 40_000),
    (5_000, 10_000), (4_000, 8_000), (2_000, 4_000), (600, 1_200), (300, 600), (80, 160), (7, 14), (3, 6),
]


SENSITIVE_10M = [9_581_728, 16_769_223]     # df values of a 10M corpus (beyond N/4) within the margin


@pytest.mark.parametrize("n,df_hi", CASES)
def test_float32_idf_is_correctly_rounded_and_insensitive(n, df_hi):
    r = pyoracle.log2_sensitivity(n, 1, df_hi, margin_ulps=4.0)
    assert r["undecidable"] == 0, r            # the 80-bit reference decides every case
    assert r["mismatch"] == 0, r               # idf == correctly rounded float32(log2(N/df))
    assert r["sensitive"] == 0, r              # ... and no last-ulp difference in math.Log could change it


def test_sensitive_document_frequencies_outside_the_benchmark_range_are_known():
    """Beyond the benchmark's df range (df > N/4) log2(N/df) approaches 0 and Log2's `Log(frac)/ln2 + exp`
    cancels (or lands near a boundary by chance): two df values of the 10M corpus come within the (4x) margin of a float32 boundary.  They are still
    correctly rounded here; the list is pinned so that a change in the restatement shows up."""
    n = 10_000_000
    r = pyoracle.log2_sensitivity(n, 2_500_001, 2 * n, margin_ulps=4.0)
    assert r["mismatch"] == 0 and r["undecidable"] == 0, r
    assert r["sensitive"] == 2, r
    found, lo = [], 2_500_001
    while True:                                   # walk the C scan from one sensitive df to the next
        r = pyoracle.log2_sensitivity(n, lo, 2 * n, margin_ulps=4.0)
        if not r["first_bad_df"]:
            break
        found.append(r["first_bad_df"])
        lo = r["first_bad_df"] + 1
    assert found == SENSITIVE_10M, found


def test_long_double_reference_agrees_with_mpmath():
    mp = pytest.importorskip("mpmath")
    mp.mp.dps = 60
    rng = np.random.default_rng(7)
    n = 10_000_000
    dfs = np.unique(np.concatenate([rng.integers(1, 2_500_001, size=400), [1, 2, 3, 1023, 1024, 1025, 2_500_000,
                                                                           9_581_728, 9_999_999, n, n + 1, 2 * n]]))
    for df in dfs.tolist():
        x = float(n) / float(df)                                   # Go rounds the quotient to float64 first
        exact = mp.log(mp.mpf(x), 2)
        want = np.float32(float(exact))                            # double rounding is safe here: checked below
        # guard the double rounding: the float64 of the exact value must not sit on a float32 midpoint
        got = np.float32(pyoracle.go_log2(x))
        assert got == want, (df, got, want)
        if x != 1.0:
            assert abs(pyoracle.go_log2(x) - float(exact)) <= 4e-16 * max(1.0, abs(float(exact))) + 3e-16, df


def test_log2_exact_powers_of_two_and_specials():
    # log10.go: Log2 returns the exponent exactly for powers of two
    for e in range(-40, 41):
        assert pyoracle.go_log2(2.0 ** e) == float(e)
    assert pyoracle.go_log2(0.0) == float("-inf")
    assert np.isnan(pyoracle.go_log2(-1.0))
    assert pyoracle.go_log2(float("inf")) == float("inf")
