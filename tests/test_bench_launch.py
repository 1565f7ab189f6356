"""bench.py's launch contract: `python bench.py --gpus N` starts its N ranks itself (no torch.distributed.run needed),
relays rank 0's JSON line, and exits non-zero when a rank fails."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    return env


def test_self_launch_reports_rank_failure():
    """Without a GPU every rank exits with an error: the parent must exit non-zero too (and print no JSON line)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("this check is for GPU-less machines")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0", "--cells", "4", "--no-cpu-baseline"],
                       env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert "needs a GPU" in r.stderr and "rank(s) failed" in r.stderr
    assert r.stdout.strip() == ""


def test_gpus_must_match_world_size():
    env = dict(_env(), WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("gpus", [2, 4])
def test_self_launched_ranks_share_one_gpu_over_gloo(gpus):
    """Plain `python bench.py --gpus N` (no launcher): N ranks on cuda:0, host-staged gloo exchange -- the code path of the
    multi-GPU run (shell -> pack -> exchange on the comm stream -> interior -> stage B) with its measurement fields."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", str(gpus), "--backend", "gloo", "--share-gpu", "--cells", "8", "--steps", "2",
                        "--warmup", "1", "--no-cpu-baseline"], env=_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == gpus and out["rccl_ranks"] == gpus and out["backend"] == "gloo" and len(out["devices"]) == gpus
    assert out["finite"] and out["value"] > 0 and out["scaling"] == "weak"
    assert out["exchange_ms"] > 0 and 0.0 <= out["overlap_frac"] <= 1.0 and out["pack_ms"] >= 0
    assert out["config"]["parallelism"] == {2: "cartesian-2x1x1", 4: "cartesian-2x2x1"}[gpus]
    # the warm-up tried the persistent grids 0 and 8 workgroups short of the chip and kept the faster setting
    assert out["reserve_cus_chosen"] in (0, 8) and set(out["reserve_cus_trial"]) == {"0", "8"}
    assert all(v["ms_per_step"] > 0 and v["exposed_exchange_ms"] >= 0 for v in out["reserve_cus_trial"].values())


@pytest.mark.gpu
def test_cfg4_on_two_ranks_and_rooflines_on_sharded_lines():
    """BASELINE configs[4] is an 8-GPU configuration: `--config cfg4 --gpus N` runs the SHARDED limited step (SubcellLimiter on a partitioned
    AderDgSolver: trace exchange + the limiter's flag and subcell-layer exchanges) with the fields of cfg 3's line, and every N > 1 line carries a
    per-rank roofline (the slowest rank's summed stage-A launches against the per-GPU work)."""
    r = subprocess.run([sys.executable, BENCH, "--config", "cfg4", "--gpus", "2", "--backend", "gloo", "--share-gpu", "--cells", "4", "--steps", "2",
                        "--warmup", "1", "--no-cpu-baseline"], env=_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["finite"] and out["value"] > 0 and out["scaling"] == "weak"
    assert out["config"]["parallelism"] == "cartesian-2x1x1" and "limiter" in out["config"]["workload"]
    assert out["exchange_ms"] > 0 and out["pack_ms"] >= 0 and 0.0 <= out["overlap_frac"] <= 1.0 and out["limiter_exchange_ms"] > 0
    rf = out["roofline"]
    assert "m8" in rf["kernel"] and rf["scope"] == "per rank (slowest)" and len(rf["launch_ms_per_rank"]) == 2
    assert rf["frac"] > 0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and rf["launches_per_step"] >= 2
    # cfg 3's sharded line: the same roofline object
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--backend", "gloo", "--share-gpu", "--cells", "6", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--no-reserve-trial"], env=_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    rf = out["roofline"]
    assert rf["bound"] == "fp64-valu" and rf["scope"] == "per rank (slowest)" and rf["frac"] > 0 and len(rf["launch_ms_per_rank"]) == 2


@pytest.mark.gpu
def test_single_gpu_line_has_roofline_and_config_variants():
    r = subprocess.run([sys.executable, BENCH, "--cells", "16", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], env=_env(),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    rf = out["roofline"]
    assert rf["bound"] == "fp64-valu" and rf["unit"] == "TFLOP/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert "traffic_source" in rf and out["dtype"] == "f64" and out["n_gpus"] == 1
    # the same configurations on the SymPy-specified Euler system (north_star: "drops in unchanged"), in the driver-run line
    oc = out["other_configs"]
    for name in ("cfg2_sympy", "cfg4_sympy"):
        assert "error" not in oc[name], oc[name]
        assert oc[name]["finite"] and oc[name]["value"] > 0 and "UserPDE" in oc[name]["roofline"]["kernel"] and oc[name]["vs_builtin_term_set"] > 0.5
        assert "SymPy" in oc[name]["term_set"]
    assert oc["cfg4"]["roofline"]["bound"] == "fp64 (valu+mfma)"
    for cfg, bound in (("cfg1", "hbm"), ("fv-ref", "hbm")):
        r = subprocess.run([sys.executable, BENCH, "--config", cfg, "--steps", "3", "--warmup", "1", "--no-cpu-baseline"], env=_env(),
                           capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        o = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
        assert o["roofline"]["bound"] == bound and o["finite"] and o["value"] > 0


@pytest.mark.gpu
def test_self_exchange_rehearsal_runs_the_sharded_step_over_rccl():
    """`--self-exchange`: one rank on the nccl (= RCCL) backend, its own neighbour in every direction -- the sharded step with its
    measurement fields over the real transport.  stdout must hold the JSON line and nothing else (RCCL prints a version table when a
    communicator comes up: bench.py keeps it off file descriptor 1)."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--self-exchange", "--cells", "8", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline"], env=_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = r.stdout.splitlines()
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["rccl_ranks"] == 1 and out["backend"] == "nccl" and "rehearsal" in out
    assert out["finite"] and out["exchange_ms"] > 0 and 0.0 <= out["overlap_frac"] <= 1.0
    assert out["roofline"]["launches_per_step"] >= 2           # shell boxes + interior box
