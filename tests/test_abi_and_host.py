"""CPU: the C-ABI library loads and exports every symbol include/ofdm_mi355x.h declares (no compute
without a GPU), the product fails loudly without a GPU, and the host-side helpers (bit packing,
frame geometry, sweep sharding) behave."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "ofdm_mi355x.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ofdm_[A-Za-z0-9_]+)\s*\(", txt)))


def test_header_symbols_are_exported_and_bound():
    from ofdm_course_amd import _lib
    lib = _lib.load()
    syms = _declared_symbols()
    assert len(syms) >= 40
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/ofdm_mi355x.h but not exported"
    # the ctypes layer binds exactly the header's entry points
    assert sorted(_lib.SIGNATURES) == syms
    assert lib.ofdm_version() >= 100


def test_no_gpu_means_loud_failure_not_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import ofdm_course_amd as ofdm
    with pytest.raises(ofdm.OfdmError) as e:
        ofdm.OFDM_demodulator(np.zeros((72, 2), complex), 8)
    assert "no CPU fallback" in str(e.value) or "HIP" in str(e.value)


def test_constellation_table_is_host_only_and_matches_oracle():
    """ofdm_constellation_func touches no GPU: usable for the layout helpers on any host."""
    import ofdm_course_amd as ofdm
    from oracle import ofdm_oracle as o
    for name in ["BPSK", "QPSK", "8PSK", "16QAM", "64QAM", "256QAM"]:
        d, bps = ofdm.constellation_func(name)
        dw, bw = o.constellation_func(name)
        assert bps == bw and np.max(np.abs(d - dw)) < 1e-15
    with pytest.raises(ofdm.OfdmError):
        ofdm.constellation_func("nope")


def test_product_code_never_imports_the_oracle():
    for base in ("ofdm-course_amd", "ofdm_course_amd"):
        for dp, _, fns in os.walk(os.path.join(ROOT, base)):
            for fn in fns:
                if fn.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                    txt = open(os.path.join(dp, fn), errors="ignore").read()
                    assert "import oracle" not in txt and "from oracle" not in txt, os.path.join(dp, fn)


def test_shipped_library_has_no_work_skipping_switches():
    """the ablation switches that skip work (OFDM_WAVE_ABL, OFDM_CHAIN_SKIP) are not in the product
    library (the former lives in the -DOFDM_DIAG build of tools/ only), and bench.py refuses to run with one set."""
    import subprocess
    import sys
    blob = open(os.path.join(ROOT, "ofdm-course_amd", "libofdm_mi355x.so"), "rb").read()
    for name in (b"OFDM_WAVE_ABL", b"OFDM_CHAIN_SKIP"):
        assert blob.count(name) == 0, name
    for var in ("OFDM_WAVE_ABL", "OFDM_CHAIN_SKIP"):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1"], capture_output=True, text=True,
                           env=dict(os.environ, **{var: "1"}))
        assert r.returncode != 0 and var in r.stderr and not r.stdout.strip(), (r.returncode, r.stderr[-300:])


def test_bit_packing_roundtrip_and_layout():
    from ofdm_course_amd import frames as fr
    rng = np.random.default_rng(0)
    for n in (1, 7, 32, 33, 2304 * 14):
        b = rng.integers(0, 2, (3, n)).astype(np.uint8)
        p = fr.pack_bits(b)
        assert p.shape == (3, 4 * ((n + 31) // 32)) and np.array_equal(fr.unpack_bits(p, n), b)
    one = np.zeros((1, 16), np.uint8); one[0, 0] = 1; one[0, 9] = 1
    assert list(fr.pack_bits(one)[0]) == [0x80, 0x40, 0, 0]          # MSB-first inside each byte


def test_frame_configs():
    from ofdm_course_amd import frames as fr
    m = fr.config_M()
    assert (m.Nfft, m.T_guard, m.N_carrier, m.K, len(m.pilotCarriers), len(m.dataCarriers)) == (2048, 256, 512, 128, 128, 384)
    assert m.frame_samples == 2304 * 14
    c5 = fr.config_C5()
    assert c5.taps.shape == (32, 2) and np.max(c5.taps[:, 0].real) < c5.K == 512 and len(c5.pilotCarriers) == 512


def test_sweep_sharding_is_a_partition():
    from ofdm_course_amd import sweep
    for world in (1, 2, 3, 4, 8):
        seen = []
        for r in range(world):
            seen += sweep.tiles_for_rank(20, 13, r, world)
        assert sorted(seen) == sorted(sweep.flatten_tiles(20, 13))          # every tile exactly once
        sizes = [len(sweep.tiles_for_rank(20, 13, r, world)) for r in range(world)]
        assert max(sizes) - min(sizes) <= 1                                 # balanced (20 SNRs vs 8 GPUs)
    assert sweep.frame_range_for_rank(8192, 3) == (3 * 8192, 4 * 8192)
    # a tile's RNG stream never depends on the rank
    assert sweep.tile_seed_stream(7, 3, 5, 64) == sweep.tile_seed_stream(7, 3, 5, 64)
    with pytest.raises(ValueError):
        sweep.tiles_for_rank(20, 13, 8, 8)


def test_mex_gateways_typecheck_against_the_header():
    """Every MEX gateway compiles (syntax/type level) against include/ofdm_mi355x.h."""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "check_mex_syntax.sh")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    names = sorted(f[:-4] for f in os.listdir(os.path.join(ROOT, "ofdm-course_amd", "mex")) if f.endswith(".cpp"))
    assert len(names) == 32 and "ofdm_rx_chain_task4" in names and "ofdm_rx_chain_task5" in names and "ofdm_task5_part2_tile" in names and "ofdm_task5_mse_tile" in names and "OMP_estimate" in names and "AutoCorrFunction" in names and "calculate_window_PAPR" in names
