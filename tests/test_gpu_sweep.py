"""The sharded BER(SNR) sweep (drivers/sweep_ber.py): two gloo ranks sharing the one GPU of the test box must
reproduce the single-process table exactly (integer counters, tile-keyed noise streams)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(cmd, out):
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    with open(out) as f:
        return json.load(f)


@pytest.mark.parametrize("config,estimator", [("M", "omp"), ("C5", "omp"), ("M", "mmse")])
def test_two_rank_sweep_equals_single_process(tmp_path, config, estimator):
    common = ["--config", config, "--estimator", estimator, "--batches", "2", "--frames-per-tile", "3",
              "--snrs", "4", "16", "28"]
    one = _run([sys.executable, "-m", "ofdm_course_amd.drivers.sweep_ber", *common, "--json", str(tmp_path / "one.json")],
               tmp_path / "one.json")
    two = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                "127.0.0.1", "--master-port", str(_free_port()), "-m", "ofdm_course_amd.drivers.sweep_ber", *common,
                "--backend", "gloo", "--force-device", "0", "--json", str(tmp_path / "two.json")], tmp_path / "two.json")
    assert two["n_gpus"] == 2 and one["n_gpus"] == 1
    assert one["errors"] == two["errors"] and one["bits"] == two["bits"]
    assert one["BER"][0] > one["BER"][-1] >= 0.0
    assert sum(one["bits"]) > 0
