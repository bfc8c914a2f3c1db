"""ofdm_task5_part2_tile (one device-resident pass per scenario of T5/Task5_part2.m: LS / MMSE / MP / OMP estimates, NMSE sums and
four BER counters for every channel realisation) against the call-by-call replay of the same driver on the oracle
(tests/oracle_lib.py) and on the per-function entries of the library."""
import numpy as np
import pytest

from oracle_lib import OracleLib

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def drv(ofdm):
    from ofdm_course_amd.drivers import task5_part2
    return task5_part2


KW = dict(Nfft=512, N_carrier=128, monteCarloRuns=5, SamplingRate=2e7, seed=5)


@pytest.mark.parametrize("profile", ["EPA", "ETU"])
def test_tile_equals_oracle_replay_fp64(ofdm, oracle, drv, profile):
    """Parity mode, comb pilots (Task5_part2.m:48-56): bit errors of all four estimators identical to the oracle replay,
    NMSE sums to 1e-9.  EPA = 7 paths (register-resident OMP), ETU = 9 paths (one-frame-per-wavefront OMP).  The scenarios keep
    Np >= dominant_taps: below that OMP_estimate.m:17 is a pinv of a rank-deficient system and its picks follow rounding noise
    in the reference itself (MP stays comparable there: test_tile_mp_with_fewer_pilots_than_paths)."""
    kw = dict(KW, combs=[4, 8, 16] if profile == "EPA" else [4, 8], DelayProfile=profile)
    got = drv.run(ofdm, batched=True, precision="fp64", **kw)
    want = drv.run(OracleLib(oracle), **kw)
    assert np.array_equal(got["_sums"]["runs"], want["_sums"]["runs"]) and np.all(got["_sums"]["runs"] == 5)
    assert np.array_equal(got["_sums"]["errors"], want["_sums"]["errors"]), (got["_sums"]["errors"], want["_sums"]["errors"])
    assert np.allclose(got["_sums"]["nmse"], want["_sums"]["nmse"], rtol=1e-9, atol=1e-12)
    assert np.array_equal(got["_sums"]["bits"], want["_sums"]["bits"])


def test_tile_equals_call_by_call_library_fp64(ofdm, drv):
    kw = dict(KW, combs=[4, 16])
    a = drv.run(ofdm, batched=True, precision="fp64", **kw)
    b = drv.run(ofdm, batched=False, **kw)
    assert np.array_equal(a["_sums"]["errors"], b["_sums"]["errors"])
    assert np.allclose(a["_sums"]["nmse"], b["_sums"]["nmse"], rtol=1e-9, atol=1e-12)


def test_tile_random_pilot_masks_fp32_mfma(ofdm, oracle, drv, monkeypatch):
    """Throughput mode on random pilot masks (Task5_part2.m:58-64, dictionary = all Nfft delays): the MP projections run on the
    matrix cores (Np = 16, 32: multiples of 16), the OMP correlation too.  Against the oracle replay: NMSE to 1e-3 relative
    (+ 1e-6 absolute), bit errors within 0.5 % of the scenario's bits; and the matrix-core MP equals the scalar fp32 MP."""
    kw = dict(KW, reg_pilot=0, Nps=[16, 32], seed=7)          # seed 7: both masks have pilot_step != 1 (data carriers exist)
    got = drv.run(ofdm, batched=True, precision="fp32", **kw)
    want = drv.run(OracleLib(oracle), **kw)
    bits = want["_sums"]["bits"].astype(float)
    assert np.all(np.abs(got["_sums"]["errors"] - want["_sums"]["errors"]) <= 0.005 * bits[None, :] + 8)
    assert np.allclose(got["_sums"]["nmse"], want["_sums"]["nmse"], rtol=2e-3, atol=1e-6)
    monkeypatch.setenv("OFDM_MP_NO_MFMA", "1")
    scal = drv.run(ofdm, batched=True, precision="fp32", **kw)
    assert np.all(np.abs(scal["_sums"]["errors"][2] - got["_sums"]["errors"][2]) <= 8)
    assert np.allclose(scal["_sums"]["nmse"][2], got["_sums"]["nmse"][2], rtol=1e-3, atol=1e-6)


def test_tile_sharded_over_two_ranks_adds_up(ofdm, drv):
    """The (kk, jj) pairs dealt round-robin over ranks (sweep.tiles_for_rank's rule): the two ranks' sums add up to the
    single-rank sums exactly (integer counters) / to rounding (NMSE)."""
    kw = dict(KW, combs=[4, 8])
    one = drv.run(ofdm, batched=True, **kw)
    parts = [drv.run(ofdm, batched=True, rank=r, world=2, **kw) for r in (0, 1)]
    assert np.array_equal(parts[0]["_sums"]["errors"] + parts[1]["_sums"]["errors"], one["_sums"]["errors"])
    assert np.array_equal(parts[0]["_sums"]["runs"] + parts[1]["_sums"]["runs"], one["_sums"]["runs"])
    assert np.allclose(parts[0]["_sums"]["nmse"] + parts[1]["_sums"]["nmse"], one["_sums"]["nmse"], rtol=1e-12)


def test_tile_mp_with_fewer_pilots_than_paths(ofdm, oracle, drv):
    """ETU has 9 paths; comb 16 leaves 8 pilots.  MP_estimate.m:10 searches Np = 8 columns for 9 iterations: the ninth finds
    every projection at -100, `max` returns column 1 again and :28-30 overwrite its coefficient.  LS, MMSE and MP rows equal
    the oracle replay (OMP is the rank-deficient pinv, not compared)."""
    kw = dict(KW, combs=[16], DelayProfile="ETU")
    got = drv.run(ofdm, batched=True, precision="fp64", **kw)
    want = drv.run(OracleLib(oracle), **kw)
    assert np.array_equal(got["_sums"]["errors"][:3], want["_sums"]["errors"][:3])
    assert np.allclose(got["_sums"]["nmse"][:3], want["_sums"]["nmse"][:3], rtol=1e-9, atol=1e-12)
