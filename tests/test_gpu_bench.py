"""bench.py as the driver runs it: one JSON line with the contract's keys at N = 1, and the N > 1 code
path (torch.distributed.run, one rank per process) rehearsed with two ranks sharing the one card over
gloo -- that run says nothing about scaling, it only keeps the multi-rank path from rotting."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

KEYS = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
        "dtype", "data", "config", "roofline"]


def _json_line(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def test_bench_single_gpu_line(gpu_ok):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "3", "--cpu-iters", "5", "--cpu-seconds", "1"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _json_line(r.stdout)
    for k in KEYS + ["cpu_baseline"]:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 3 and d["higher_is_better"] is True
    assert d["unit"] == "Mpix*iter/s" and d["dtype"] == "f32" and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["config"]["workload"].startswith("1920x1080") and "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    assert rf["traffic"] and rf["traffic"] < rf["algorithmic_bytes_per_launch"]  # the committed PMC summary of this workload
    assert abs(d["value"] - 1920 * 1080 * 100 / (d["ms_per_step"] * 1e-3) / 1e6) < 1e-3 * d["value"]
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and cb["unit"] == "Mpix*iter/s"


def test_bench_two_ranks_rehearsal(gpu_ok):
    env = dict(os.environ, HSFLOW_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29517", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "3"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    d = _json_line(r.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and "cpu_baseline" not in d
    assert abs(d["value"] - 2 * 1920 * 1080 * 100 / (d["ms_per_step"] * 1e-3) / 1e6) < 1e-3 * d["value"]


@pytest.mark.parametrize("extra", [[], ["--overlap"]])
def test_slab_two_ranks_rehearsal_is_bit_identical(gpu_ok, extra):
    """tools/bench_slab.py with two ranks sharing the card (gloo, halo rows staged through the host): the
    enqueue-only driver (context on torch's stream, no host waits between chunks) must give the whole-frame
    result bit for bit on the rows each rank owns."""
    env = dict(os.environ, HSFLOW_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29519", os.path.join(ROOT, "tools", "bench_slab.py"), "--width", "1500", "--height", "700",
                        "--iters", "53", "--halo", "12", "--steps", "1", "--check"] + extra,
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    assert r.stdout.count("identical to the whole-frame solve") == 2, r.stdout
    d = _json_line(r.stdout)
    assert d["n_gpus"] == 2 and d["exchanges"] == 4


def test_graft_entry_smoke(gpu_ok):
    """__graft_entry__.smoke() as the driver calls it (fresh interpreter, cuda:0)."""
    r = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.smoke(); print('SMOKE-OK')"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0 and "SMOKE-OK" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])
