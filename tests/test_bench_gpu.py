"""bench.py itself under the driver's GPU tests: the RCCL code path (a single-rank process group) of both workloads and
the JSON contract of the default line with the north-star-loop sub-objects, so that a SCALE run is never the first
execution of that code."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
        "vs_baseline", "dtype", "data", "config", "roofline")


def _bench(args, env=None, timeout=900):
    e = dict(os.environ)
    e.update(env or {})
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True,
                         timeout=timeout, env=e, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    line = res.stdout.strip().splitlines()[-1]
    return json.loads(line)


@pytest.mark.parametrize("workload,port", [("solve", "29531"), ("config5", "29532")])
def test_rccl_path_single_rank(workload, port):
    out = _bench(["--main-only", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--workload", workload,
                  "--batch", "65536"], env={"ANET_BENCH_FORCE_DIST": "1", "MASTER_PORT": port})
    for k in KEYS:
        assert k in out, k
    assert "allgather(costs)" in out["config"]["parallelism"]
    assert out["n_gpus"] == 1 and out["steps"] == 3 and out["value"] > 0
    # the line describes its collective: ranks the RCCL group really has, the all-gather alone, its payload
    cfg = out["config"]
    assert cfg["ranks_seen"] == 1 and cfg["allgather_ms"] > 0 and cfg["allgather_every"] == 1
    assert cfg["allgather_bytes_per_rank"] == 8 * (65536 if workload == "solve" else 32768)
    assert out["roofline"]["bound"] == ("hbm" if workload == "solve" else "fp64")
    assert 0 < out["roofline"]["frac"] < 1


def test_allgather_every_k_skips_collectives():
    out = _bench(["--main-only", "--steps", "6", "--warmup", "0", "--no-cpu-baseline", "--batch", "65536", "--allgather-every", "3"],
                 env={"ANET_BENCH_FORCE_DIST": "1", "MASTER_PORT": "29533"})
    assert out["config"]["allgather_every"] == 3 and out["config"]["allgathers_in_timed_loop_and_warmup"] == 2


def test_self_launch_path():
    """`python bench.py --gpus N` starts its own ranks: the same code path forced with one rank (ANET_BENCH_SELF_LAUNCH=1
    re-executes under torch.distributed.run), RCCL group included; the JSON line stays the last line of stdout."""
    for workload in ("solve", "config5"):
        out = _bench(["--gpus", "1", "--main-only", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--workload", workload,
                      "--batch", "65536"], env={"ANET_BENCH_SELF_LAUNCH": "1", "ANET_BENCH_FORCE_DIST": "1"})
        for k in KEYS:
            assert k in out, k
        assert out["n_gpus"] == 1 and out["steps"] == 3 and out["value"] > 0
        assert "allgather(costs)" in out["config"]["parallelism"]


def test_gpus_beyond_the_box_is_clamped_not_refused():
    """--gpus 8 on a box with fewer GPUs runs the ranks there are and says so (n_gpus = what ran)."""
    import torch
    n = torch.cuda.device_count()
    out = _bench(["--gpus", "8", "--main-only", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--batch", "65536"])
    assert out["n_gpus"] == min(8, n) and out["value"] > 0


def test_default_line_carries_the_north_star_loop():
    """configs[1] headline + config3 / config4 / config5 / qp_solve sub-objects, each with a roofline; the line fits the
    driver's tail as printed; the whole-batch gradient parity figures of the config3 leg are at rounding level (round 4's
    driver line carried a false 4e-6: the kernel split had overwritten the buffer); short CPU legs."""
    e = dict(os.environ)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--cpu-seconds", "2",
                          "--batch", "65536"], capture_output=True, text=True, timeout=1200, env=e, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    line = res.stdout.strip().splitlines()[-1]
    assert len(line) <= 7000, len(line)
    out = json.loads(line)
    assert "dropped" not in out
    for k in KEYS:
        assert k in out, k
    c3, c4, c5 = out["config3"], out["config4"], out["config5"]
    for leg in (c3["b4096"], c3["saturating"], c4, c5):
        r = leg["roofline"]
        assert r["bound"] == "fp64" and 0 < r["frac"] < 1 and 0 < r["hbm"]["frac"] < 1
    assert c3["b4096"]["batch"] == 4096 and c4["batch"] == 4096
    cb = c3["cpu_baseline"]
    assert cb["gpu_vs_cpu_max_rel_cost_err"] <= 1e-9
    assert cb["gpu_vs_cpu_max_rel_gradT_err"] <= 1e-7 and cb["gpu_vs_cpu_max_rel_gradP_err"] <= 1e-7
    # every problem of configs[3] stopped on its own (LBFGS_CONVERGENCE = 0 / LBFGS_STOP = 1)
    assert set(c4["status_hist"]) <= {"0", "1"} and sum(c4["status_hist"].values()) == 4096
    assert c4["evals_max"] < c4["max_evals_cap"]
    assert out["config"]["ranks_seen"] == 1 and out["qp_solve"]["snap8"]["batch"] == 4096
    b1 = out["config1_b1024"]
    assert b1["stream_ms_min_median_max"][0] <= b1["stream_ms_per_step"] <= b1["stream_ms_min_median_max"][2]


@pytest.mark.parametrize("leg,key", [("config3", "config3"), ("qp", "qp_solve")])
def test_one_leg_alone_is_a_contract_line(leg, key):
    """`--workload config3|qp --main-only`: the form tools/profile_leg.sh runs under rocprofv3 -- the leg's own generator and
    timing, nothing else launched beside it; the line names the kernel that ran and carries the committed PMC traffic."""
    out = _bench(["--workload", leg, "--main-only", "--no-cpu-baseline"])
    for k in KEYS:
        assert k in out, k
    assert out["config"]["leg"] == leg and out["config"]["main_only"] is True and key in out and out["value"] > 0
    if leg == "config3":
        b, sat = out["config3"]["b4096"], out["config3"]["saturating"]
        assert b["launches_per_step"] == 1 and b["roofline"]["kernel"] == "k_minco_cost_grad_fused"
        assert sat["launches_per_step"] == 3 and sat["roofline"]["kernel"].startswith("k_piece_grad")
        assert "kernel_split_us" not in b and "three_launch_split_us" not in b        # no split under --main-only
        # (the main roofline is the leg's own plus the contract's "peak" / "unit", which a sub-leg states once per line)
        assert out["roofline"] == dict(b["roofline"], peak=out["roofline"]["peak"], unit="TFLOP/s")
        assert abs(out["ms_per_step"] - b["ms_per_step"]) < 1e-9
        for r in (b["roofline"], sat["roofline"]):
            if r["traffic"] is not None:      # (a committed profile of this leg; the 25 MB of the literal batch partly stay in the L2s from launch to launch: < 1)
                assert 0.5 < r["hbm"]["traffic_over_algorithmic"] < 4.0


def test_launch_shapes_follow_the_devices_compute_units():
    """anet_compute_units = the device's multiprocessor count; the one-launch / three-launch decision of the cost + gradient
    evaluation is a number of rounds of workgroups per compute unit (three rounds of groups of 16 for <= 8 pieces)."""
    import torch
    import allocnet_amd as aa
    ctx = aa.Context(0)
    cus = ctx.compute_units
    assert cus == torch.cuda.get_device_properties(0).multi_processor_count and cus > 0
    pen = aa.make_penalty(rho=50.0, w_corridor=1e4, w_vel=1e3, w_acc=1e3, smooth_mu=1e-2, max_vel=4.0, max_acc=6.0, res=20, poly_rows=16)
    # (8-piece snap with c = 3 at 20 samples: phase 2 on the matrix instructions and eight waves -- six rounds; any other boundary
    #  count or sample count: the vector phase 2, three rounds)
    mx = os.environ.get("ANET_FUSED_MX", "1") != "0"
    assert aa.minco_cost_grad_launches(4, 8, 16 * (6 if mx else 3) * cus, penalty=pen, ctx=ctx) == 1
    assert aa.minco_cost_grad_launches(4, 8, 16 * (6 if mx else 3) * cus + 1, penalty=pen, ctx=ctx) == 3
    assert aa.minco_cost_grad_launches(4, 8, 16 * 3 * cus, penalty=pen, ctx=ctx, c=4) == 1
    assert aa.minco_cost_grad_launches(4, 8, 16 * 3 * cus + 1, penalty=pen, ctx=ctx, c=4) == 3
    assert aa.minco_cost_grad_launches(3, 16, 8 * (8 if mx else 2) * cus, penalty=pen, ctx=ctx) == 1
    assert aa.minco_cost_grad_launches(3, 16, 8 * (8 if mx else 2) * cus + 1, penalty=pen, ctx=ctx) == 3
    assert aa.minco_cost_grad_launches(4, 8, 64, penalty=None, ctx=ctx) == 3          # no penalty: the streaming kernels
    # the penalty kernel's launch shape: the small-batch shapes by pairs per compute unit, above them the matrix-instruction
    # kernel for orders 3 / 4 at 20 samples per piece (k_piece_grad_mx), the lane-per-pair kernel otherwise
    big = 64 * cus + 1
    assert aa.minco_piece_grad_shape(4, 8, big, penalty=pen, ctx=ctx) == (3 if os.environ.get("ANET_PG_MX", "1") != "0" else 0)
    assert aa.minco_piece_grad_shape(3, 16, big, penalty=pen, ctx=ctx) == (3 if os.environ.get("ANET_PG_MX", "1") != "0" else 0)
    pen9 = aa.make_penalty(rho=50.0, w_corridor=1e4, w_vel=1e3, w_acc=1e3, smooth_mu=1e-2, max_vel=4.0, max_acc=6.0, res=9, poly_rows=16)
    assert aa.minco_piece_grad_shape(4, 8, big, penalty=pen9, ctx=ctx) == 0
    assert aa.minco_piece_grad_shape(2, 8, big, penalty=pen, ctx=ctx) == 0
    assert aa.minco_piece_grad_shape(4, 8, 4 * cus, penalty=pen, ctx=ctx) == 2 and aa.minco_piece_grad_shape(4, 8, 32 * cus, penalty=pen, ctx=ctx) == 1
    assert aa.minco_cost_grad_launches(2, 8, 64, penalty=pen, ctx=ctx, c=2) == 3      # order 2 has no one-launch instantiation
