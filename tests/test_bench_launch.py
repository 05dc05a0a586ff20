"""CPU: bench.py starts its own ranks when asked for N > 1 GPUs without a launcher (child process of
``python -m torch.distributed.run``, started before anything touches the GPU), refuses a WORLD_SIZE that contradicts
--gpus, and the BASELINE config presets resolve.  ``--dry-run`` = rank start-up + barrier only (gloo here)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["RATO_DIST_BACKEND"] = "gloo"
    return env


def test_bench_spawns_its_own_ranks():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run"], env=_env(), capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout                      # ONE JSON line, from rank 0
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["dry_run"] is True


def test_bench_refuses_contradicting_world_size():
    env = _env()
    env.update(WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run"], env=env, capture_output=True, text=True,
                         timeout=120)
    assert out.returncode == 2 and "WORLD_SIZE=3" in out.stderr


def test_config_presets():
    sys.path.insert(0, ROOT)
    import bench
    argv = sys.argv
    try:
        for cfg, want in (("metric", ("drone", 100000, 50)), ("C2", ("drone", 10000, 50)),
                          ("C3", ("driving", 10000, 40)), ("C4", ("hopper", 50000, 60)), ("C5", ("driving", 125000, 40))):
            sys.argv = ["bench.py", "--config", cfg]
            a = bench.parse()
            assert (a.workload, a.M, a.S) == want
        sys.argv = ["bench.py", "--config", "C5", "--M", "1000"]
        a = bench.parse()
        assert (a.workload, a.M, a.S) == ("driving", 1000, 40)
    finally:
        sys.argv = argv


def test_bench_eight_ranks_dry_run_c5():
    """the launch path of `bench.py --gpus 8 --config C5` (BASELINE C5: driving, M = 1e6 over 8 GPUs): eight ranks come
    up (gloo here), LOCAL_RANK -> device, 125,000 samples per GPU, M_total = 1,000,000, ONE line from rank 0"""
    out = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--config", "C5", "--dry-run", "--strict-comm"],
                         env=_env(), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    line = json.loads(lines[0])
    cfg = line["config"]
    assert line["n_gpus"] == 8 and line["scaling"] == "weak"
    assert (cfg["workload"], cfg["M_per_gpu"], cfg["S"], cfg["M_total"]) == ("driving", 125000, 40, 1000000)
    assert [r["device"] for r in sorted(cfg["ranks"], key=lambda r: r["rank"])] == [f"cuda:{i}" for i in range(8)]
    assert cfg["strict_comm"] is True
