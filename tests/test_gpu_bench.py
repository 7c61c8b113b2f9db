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
    for k in KEYS + ["cpu_baseline", "ms_per_step_min", "ms_per_step_median", "ms_per_step_max", "other_termination"]:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 3 and d["higher_is_better"] is True
    assert d["unit"] == "Mpix*iter/s" and d["dtype"] == "f32" and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert "1920x1080" in d["config"]["workload"] and "model" not in d["config"]
    # the headline is the reference's own call form (ITER|EPS, eps 1e-6) on a NEW resident pair every step (its camera loop,
    # OpticalFlowOpenCV.cpp:91-95) through the device-resident pair pipeline; ITER through the same loop beside it; no early
    # stop on these pairs; every pair's own early-stop check is settled (nothing carried over between steps)
    assert d["config"]["termination"].startswith("ITER|EPS") and d["config"]["iterations_done"] == 100 and d["config"]["eps_rerun"] == 0
    assert d["config"]["loop"] == "stream" and d["config"]["call"].startswith("hsflow_pipeline_submit_device, 6 slots on 2 streams") and "own check" in d["config"]["eps_check"]
    assert d["other_termination"]["termination"] == "ITER" and d["other_termination"]["ms_per_step"] > 0
    assert d["fresh_frames"]["is_the_headline"] is True and d["fresh_frames"]["ms_per_step"] == d["ms_per_step"]
    # rounds 1-2's loop (the same pair again and again on one context) beside it; the stream must not be slower than 1.03 x that ITER figure
    sc = d["single_context"]
    assert sc["ITER"]["ms_per_step"] > 0 and sc["ITER|EPS (eps 1e-6)"]["ms_per_step"] > 0
    assert d["ms_per_step"] <= 1.03 * sc["ITER"]["ms_per_step"], (d["ms_per_step"], sc)
    # the reference's OpenCL discretisation on the same frames, as a side figure (register-strip kernel at this size)
    assert d["classic_mode"]["kernel"] == "strip" and 0 < d["classic_mode"]["ms_per_step"] < 5
    assert d["ms_per_step_min"] <= d["ms_per_step_median"] <= d["ms_per_step_max"] and len(d["ms_per_step_blocks"]) == 5
    assert d["ms_per_step_blocks"][0] == d["ms_per_step"]
    rf = d["roofline"]
    # the bound that binds the multi-sweep kernel is VALU issue: a fraction in (0, 1], recomputable from its parts
    assert rf["bound"] == "valu" and rf["unit"] == "Tlane-op/s" and 0 < rf["frac"] <= 1
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    vi = rf["valu_issue"]
    c = vi["constants"]
    peak = c["simds"] * 64.0 / c["cycles_per_wave64_op_measured"] * c["clock_ghz_measured"] * 1e9 / 1e12
    assert abs(rf["peak"] - peak) < 1e-9 * peak
    ach = vi["op_slots_per_pixel_sweep"] * 1920 * 1080 * rf["sweeps_per_launch"] / (rf["avg_launch_us"] * 1e-6) / 1e12
    assert abs(rf["achieved"] - ach) < 1e-6 * ach
    assert abs(vi["ideal_us_per_step"] / vi["jacobi_kernel_us_per_step"] - rf["frac"]) < 1e-6
    assert abs(vi["frac_at_step_rate"] - vi["ideal_us_per_step"] / (d["ms_per_step"] * 1e3)) < 1e-9
    assert rf["traffic_measured_in_run"] is False and (rf["traffic"] is None or rf["traffic"] < rf["hbm_algorithmic"]["bytes_per_launch"])
    assert rf["hbm_algorithmic"]["bytes_per_pixel_sweep"] == 28.0
    assert abs(d["value"] - 1920 * 1080 * 100 / (d["ms_per_step"] * 1e-3) / 1e6) < 1e-3 * d["value"]
    # the workload the reference's authors ran (600x480 city pair, lambda 0.1, 100 iterations, blur first): latency, stream, end to end, CPU port
    rd = d["reference_default_workload"]
    assert "error" not in rd, rd
    assert rd["latency"]["iterations_done"] == 100 and 0 < rd["stream_shape_by_hand"]["ms"] and 0 < rd["stream"]["ms"] and 0 < rd["end_to_end"]["ms"]
    assert rd["stream"]["tiles"] == rd["stream_shape_by_hand"]["tiles"] and rd["stream"]["rows"] == 5   # the pipeline picks that shape itself
    assert rd["latency"]["ms"] < 0.2 and rd["cpu_port"]["cores"] == 1 and rd["cpu_port"]["ms"] > rd["latency"]["ms"]
    assert 0 < d["classic_mode"]["stream_ms_per_step"] < d["classic_mode"]["ms_per_step"] * 1.05
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and cb["unit"] == "Mpix*iter/s"


def test_bench_two_ranks_rehearsal(gpu_ok):
    """The N > 1 line as the driver's SCALE run would produce it, rehearsed with two ranks sharing the one card
    over gloo (halo rows staged through the host): the headline plus the C4 pipeline and the C5 slab
    measurements (shrunk: 16 pairs, a 2048^2 frame, 70 sweeps), the slab result checked bit for bit."""
    env = dict(os.environ, HSFLOW_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29517", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "3",
                        "--c4-pairs", "16", "--c5-size", "2048", "--c5-iters", "70", "--c5-halo", "16"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    d = _json_line(r.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and "cpu_baseline" not in d
    assert d["rccl_ranks"] == 2 and d["backend"] == "gloo"
    assert abs(d["value"] - 2 * 1920 * 1080 * 100 / (d["ms_per_step"] * 1e-3) / 1e6) < 1e-3 * d["value"]
    c4, c5 = d["c4_pipeline"], d["c5_slab"]
    assert c4["pairs_total"] == 16 and c4["pairs_this_rank"] == 8 and c4["value"] > 0 and c4["collective"].startswith("none")
    assert c5["owned_rows_bit_identical_to_band_solve"] is True
    assert c5["plain"]["exchanges"] == 4 and c5["overlap"]["exchanges"] == 4 and c5["plain"]["value"] > 0 and c5["overlap"]["value"] > 0


def test_bench_gpus_2_launches_its_own_ranks(gpu_ok):
    """`python bench.py --gpus 2 ...` with no launcher and no WORLD_SIZE (how the driver starts the N = 1 run; if the
    SCALE run is started the same way it must not die before measuring): bench.py starts the two ranks itself and
    relays rank 0's line."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["HSFLOW_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "3",
                        "--c4-pairs", "16", "--c5-size", "2048", "--c5-iters", "70", "--c5-halo", "16"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    d = _json_line(r.stdout)
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["backend"] == "gloo"
    assert "c4_pipeline" in d and "c5_slab" in d and d["c5_slab"]["owned_rows_bit_identical_to_band_solve"] is True
    for k in KEYS:
        assert k in d, k


def test_bench_ranks_give_up_on_a_hung_side_measurement(gpu_ok):
    """A C5 exchange that never returns (it has never run over RCCL on real peers) must not cost the SCALE record its
    headline: after --side-timeout every rank ends itself, rank 0 having printed the line with the error under the key."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["HSFLOW_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "3", "--no-side",
                        "--c4-pairs", "8", "--c5-size", "1024", "--c5-iters", "20", "--side-timeout", "20", "--debug-hang", "c5"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    d = _json_line(r.stdout)
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["c4_pipeline"]["value"] > 0
    assert "abandoned" in d["c5_slab"]["error"]


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
