"""GPU: the driver's command line end to end at a small size — `python bench.py` with its default blocks (products,
factored and regenerated-noise regions, SCP block, CPU baseline) prints exactly ONE JSON line carrying every block."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, capture_output=True, text=True,
                         timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


def test_default_blocks_at_a_small_size():
    d = _run(["--M", "3000", "--S", "20", "--steps", "3", "--warmup", "1", "--scp-iters", "3"])
    assert d["metric"] == "SAA constraint-eval throughput" and d["n_gpus"] == 1 and d["steps"] == 3
    assert d["value"] > 0 and d["value_factored"] > 0 and d["value_regenerated"] > 0
    for k in ("roofline", "roofline_factored", "roofline_regenerated"):
        r = d[k]
        assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["achieved"] > 0 and 0 < r["frac"] < 1
    assert d["roofline_regenerated"]["algorithmic_bytes_per_launch"] < d["roofline"]["algorithmic_bytes_per_launch"]
    assert d["scp"]["iters"] == 3 and d["scp"]["cumulative_s"] > 0
    kkt = d["scp"]["kkt"]                               # matrix-free certificate against the reference-layout QP
    assert "error" not in kkt, kkt
    assert max(kkt["primal"], kkt["stationarity"], kkt["complementarity"]) < 1e-7, kkt
    assert d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["kind"] == "port"
    cb = d["cpu_baseline"]                              # timed at 1 / 64 / half / all threads; the best leg is the value
    assert cb["cores"] >= 1 and cb["value"] == max(cb["value_by_threads"].values()) and cb["cpus_available"] >= cb["cores"]
    # the whole metric inside `config` (what the driver's record keeps): the SCP wall-clock next to the throughput
    cfg = d["config"]
    assert cfg["scp_cumulative_s"] == d["scp"]["cumulative_s"] and cfg["scp_iters"] == 3
    assert cfg["scp_cuts_total"] == d["scp"]["cuts_total"] and cfg["gpu_over_cpu"] == cb["gpu_over_cpu"]
    assert list(d)[-1] == "config" and list(d)[-2] == "roofline"      # the contract's keys close the line
    assert d["device"]["sclk_mhz_beside_hot_kernel"] > 500


@pytest.mark.parametrize("config", ["C3", "C4"])
def test_other_configs_run(config):
    d = _run(["--config", config, "--steps", "3", "--warmup", "1", "--no-cpu-baseline"])
    assert d["value"] > 0 and d["config"]["baseline_config"] == config and d["roofline"]["kernel_ms"] > 0


def test_two_ranks_validate_their_exchange_before_timing():
    """`bench.py --gpus 2` (two ranks sharing the one GPU of the test box over gloo): before the timed region the run
    validates the exchange it is about to time (dist.comm_selfcheck) and reports it with the per-rank kernel times."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(RATO_DIST_BACKEND="gloo", RATO_SINGLE_GPU="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--M", "3000", "--S", "20",
                          "--steps", "3", "--warmup", "1", "--jacobian", "products"], capture_output=True, text=True,
                         timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    cfg = d["config"]
    assert d["n_gpus"] == 2 and cfg["M_total"] == 6000 and d["value"] > 0
    sc = cfg["comm_selfcheck"]
    assert sc["ok"] and sc["bitwise_vs_torch_all_gather"] and sc["identical_on_every_rank"] and sc["world"] == 2
    assert cfg["rccl_ranks"] == 0 and "gloo" in cfg["transport"]          # two ranks on one device: RCCL refuses that
    assert len(cfg["kernel_ms_ranks"]["per_rank"]) == 2 and cfg["kernel_ms_ranks"]["min"] > 0
    # N > 1 without --overlap / --no-overlap: both step forms probed (slowest rank's clock), the faster one timed
    fp = cfg["step_form_probe"]
    assert fp["serial_us"] > 0 and fp["pipelined_us"] > 0
    assert fp["timed"] == ("pipelined" if fp["pipelined_us"] < fp["serial_us"] else "serial")
    # --strict-comm: a run whose exchange is not the library's RCCL communicator on every rank must not report a number
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--M", "3000", "--S", "20",
                          "--steps", "3", "--warmup", "1", "--jacobian", "products", "--strict-comm"], capture_output=True,
                         text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode != 0 and not [l for l in out.stdout.splitlines() if l.startswith("{")]
