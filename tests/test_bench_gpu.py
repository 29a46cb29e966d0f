"""bench.py itself under test on the one-GPU box: the multi-rank launch line the driver uses (torch.distributed.run, one
process per rank) with two ranks sharing GPU 0 and gloo as the transport, every guide-exchange mode; the lock-step recovery
path driven over several steps; and, when the box has two GPUs, the same over RCCL."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _run(cmd, timeout=420):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0, f"rc {r.returncode}\n{r.stdout[-2000:]}\n{r.stderr[-3000:]}"
    assert len(lines) == 1, f"expected ONE JSON line, got {len(lines)}\n{r.stdout[-2000:]}"
    return json.loads(lines[0])


def _torchrun(n, port, extra):
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), "bench.py", "--gpus", str(n)] + extra


COMMON = ["--batch", "4", "--steps", "2", "--warmup", "1", "--workload", "full", "--no-cpu-baseline", "--no-e2e"]


@pytest.fixture(scope="module")
def single_rank_frame0(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("bench1") / "f0.npz")
    res = _run([sys.executable, "bench.py", "--gpus", "1", "--dump-frame0", out] + COMMON)
    assert res["n_gpus"] == 1 and res["lockstep_timeouts"] == 0
    return np.load(out)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("exchange", ["broadcast", "scatter", "auto"])
def test_bench_two_ranks_share_gpu0_over_gloo(tmp_path, single_rank_frame0, exchange):
    """exactly the driver's multi-GPU command (minus the backend): one JSON line from rank 0, n_gpus = 2, no lock-step
    time-outs, the one-round exchange probe present, and rank 0's frame 0 bit-identical to the single-rank run"""
    out = str(tmp_path / "f0.npz")
    port = 29700 + (os.getpid() % 200) + {"broadcast": 0, "scatter": 1, "auto": 2}[exchange]
    res = _run(_torchrun(2, port, ["--dist-backend", "gloo", "--guide-exchange", exchange, "--dump-frame0", out] + COMMON))
    assert res["n_gpus"] == 2 and res["steps"] == 2 and res["scaling"] == "weak"
    assert res["lockstep_timeouts"] == 0 and res["lockstep_recomputed"] is False
    assert res["value"] > 0 and res["config"]["frames_per_step_per_gpu"] == 4
    assert res["config"]["guide_exchange"] in (("broadcast", "scatter") if exchange == "auto" else (exchange,))
    probe = res["config"]["guide_exchange_probe"]
    assert probe["broadcast_one_round_ms"] > 0 and probe["scatter_one_round_ms"] > 0
    got = np.load(out)
    assert np.array_equal(got["disp"], single_rank_frame0["disp"])
    assert np.array_equal(got["q"], single_rank_frame0["q"])


@pytest.mark.timeout(600)
def test_bench_recovers_from_lockstep_timeouts_over_several_steps(tmp_path, single_rank_frame0):
    """every lock-step workgroup of the first timed region reports a time-out (vdd_spin_limit = -1): the first step poisons
    its output and raises the host flag, the following steps are refused (V3D_ERR_LOCKSTEP) -- bench.py must neither
    traceback nor report that region: it switches the handle to per-direction launches, times again, says so, and the
    recomputed output has the bits of the healthy run"""
    out = str(tmp_path / "f0.npz")
    res = _run([sys.executable, "bench.py", "--gpus", "1", "--steps", "3", "--test-inject-lockstep-timeout", "--dump-frame0", out]
               + [a for a in COMMON if a not in ("--steps", "2")] + ["--no-cli"])
    assert res["lockstep_timeouts"] > 0 and res["lockstep_recomputed"] is True
    assert res["lockstep_timeouts_after_switch"] == 0 and res["value"] > 0
    got = np.load(out)
    assert np.array_equal(got["disp"], single_rank_frame0["disp"]) and np.array_equal(got["q"], single_rank_frame0["q"])


@pytest.mark.timeout(900)
def test_bench_two_ranks_over_rccl(tmp_path, single_rank_frame0):
    """the driver's command as it is (backend nccl = RCCL, one GPU per rank); skips on a one-GPU box"""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    out = str(tmp_path / "f0.npz")
    res = _run(_torchrun(2, 29950 + (os.getpid() % 40), ["--dump-frame0", out] + COMMON))
    assert res["n_gpus"] == 2 and res["lockstep_timeouts"] == 0
    assert res["config"]["guide_exchange"] == "scatter"
    got = np.load(out)
    assert np.array_equal(got["disp"], single_rank_frame0["disp"]) and np.array_equal(got["q"], single_rank_frame0["q"])
