"""CPU checks of bench.py's bookkeeping: algorithmic bytes, PMC traffic lookup, CLI."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_algorithmic_bytes_match_survey():
    import bench
    assert bench.algorithmic_bytes(4, 3, 8) == 1920      # 8-seg snap, PVA ends
    assert bench.algorithmic_bytes(4, 4, 8) == 1968      # MINCO ends
    assert bench.algorithmic_bytes(3, 3, 16) == 2944     # 16-seg jerk


def test_config5_workload_bookkeeping():
    """`--workload config5` / the config5 sub-object: SURVEY 8(d) bytes of one cost + gradient evaluation, the CLI
    switch, and the shard arithmetic it relies on."""
    import bench
    from allocnet_amd.distributed import shard_bounds
    assert bench.config5_bytes(4, 3, 8, 16) == 1920 + 4096 + 232 == 6248
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert '"--workload"' in src and '"config5"' in src and "scaling\": \"strong" in src.replace("'", '"')
    assert sum(shard_bounds(32768, 8, r)[1] - shard_bounds(32768, 8, r)[0] for r in range(8)) == 32768
    assert shard_bounds(32768, 3, 0) == (0, 10923) and shard_bounds(32768, 3, 2) == (21846, 32768)


def test_pmc_traffic_lookup_uses_committed_profile():
    import bench
    t = bench.pmc_traffic_bytes(1 << 20, 8, 4)
    assert t is not None
    alg = 1920 * (1 << 20)
    assert abs(t - alg) / alg < 0.02          # measured HBM bytes ~ algorithmic bytes: no wasted traffic
    assert bench.pmc_traffic_bytes(12345, 8, 4) is None
    f = os.path.join(ROOT, "profiles", "r01_pmc.json")
    d = json.load(open(f))
    assert any(e["grid"] == 1 << 20 for e in d["k_minco_solve"])


def test_cli_refuses_to_run_without_a_gpu_or_reports():
    """No GPU in the CPU suite: bench.py must exit with its explicit message, never fall back to a CPU path."""
    import torch
    if torch.cuda.is_available():
        return
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=300)
    assert res.returncode != 0
    assert "needs a GPU" in (res.stderr + res.stdout)


def test_north_star_loop_bookkeeping():
    """config3 / config4 sub-objects (BASELINE configs[2] / configs[3]): analytic FLOP counts, the FP64 roofline object's
    keys, and that the config5 line is no longer labelled HBM-bound."""
    import bench
    assert bench.classic_solve_flops(4, 8) == 2 * 64 * 8 * 8 + 3 * 2 * 64 * 16 == 14336      # 8.2 k + 6.1 k
    f = bench.cost_grad_flops(4, 8, 16, 20)
    assert f == 8 * 20 * (144 + 112 + 12) + 14336 + 6144 + 1536 == 64896
    assert bench.cost_grad_flops(3, 16, 16, 20) > f
    r = bench.fp64_roofline(74.5e12, 2.0, 8e12, "k")
    assert r["bound"] == "fp64" and abs(r["frac"] - 0.5) < 1e-12 and "traffic" in r and "peak" not in r
    assert abs(r["hbm"]["frac"] - 0.5) < 1e-12
    # (a sub-leg's roofline is a fraction of "fp64_peak_tflops", stated once per line; as a line's MAIN roofline it carries the contract's keys)
    rp = bench.with_peak(r)
    assert rp["peak"] == bench.FP64_PEAK_TFLOPS and rp["unit"] == "TFLOP/s" and rp["frac"] == r["frac"]
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'out["config3"] = run_config3' in src and 'out["config4"] = run_config4' in src
    assert '"bound": "hbm"' not in src[src.index("def run_config5"):src.index("def synth_batch_minor")]
    # the oracle is only reached from the cpu_baseline legs
    for fn in ("run_config3", "run_config4"):
        body = src[src.index(f"def {fn}"):]
        body = body[:body.index("\ndef ", 10)]
        assert body.index("from oracle import cbind") > body.index("if cpu_baseline:")


def test_qp_solve_bookkeeping():
    """qp_solve sub-object (the reference's online solve): analytic operation count of a Newton step and
    that the oracle is reached from the cpu_baseline leg only."""
    import bench
    f = bench.qp_newton_step_flops(4, 8, 16, 20)
    rows = 8 * 20 * 28
    assert f == 5 * 30 * rows + 11 * 160 * 9 * 8 * 2 + 8 * 64 * 20 * 12 * 2 + 9 * (576 + 2 * 1728) + 4 * 17 * 144
    assert 1.0e6 < f < 2.5e6                         # profiles/r03_qp_ipm_roofline.txt: ~2 MFLOP per step
    assert bench.qp_newton_step_flops(3, 5, 16, 20) < f
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'out["qp_solve"] = run_qp' in src
    body = src[src.index("def run_qp"):]
    body = body[:body.index("\ndef ", 10)]
    assert body.index("from oracle import qp_np") > body.index("if cpu_baseline and extras:")
    assert body.index("from oracle import cbind") > body.index("if cpu_baseline and extras:")
    # the like-for-like CPU figure is the structured port; the dense numpy interior point is kept beside it
    assert '"kind": "port"' in body and "qp_ipm_batch" in body and '"dense_numpy": dense_numpy' in body


def test_launch_plan_and_self_launch_command():
    """`python bench.py --gpus N`: N > 1 without WORLD_SIZE starts its own ranks (clamped to the visible GPUs, with a
    message); under torch.distributed.run the process is a rank; one GPU runs in place unless the self-launch path is forced."""
    import bench
    assert bench.launch_plan(1, {}, 1) == ("run", 1, None)
    assert bench.launch_plan(8, {}, 8) == ("spawn", 8, None)
    plan, n, msg = bench.launch_plan(8, {}, 2)
    assert (plan, n) == ("spawn", 2) and "2 GPU(s) visible" in msg
    plan, n, msg = bench.launch_plan(8, {}, 1)
    assert (plan, n) == ("run", 1) and msg
    assert bench.launch_plan(1, {"ANET_BENCH_SELF_LAUNCH": "1"}, 1) == ("spawn", 1, None)
    assert bench.launch_plan(4, {"WORLD_SIZE": "4", "RANK": "1"}, 8) == ("run", 4, None)
    plan, n, msg = bench.launch_plan(8, {"WORLD_SIZE": "2"}, 8)
    assert (plan, n) == ("run", 2) and "WORLD_SIZE=2" in msg
    cmd = bench.self_launch_cmd(2, ["--gpus", "8", "--steps", "3", "--workload", "config5"], 12345)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=2" in cmd
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "2", "--steps", "3", "--workload", "config5"]
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "12345"
    assert bench.self_launch_cmd(4, ["--gpus=8", "--main-only"], 1)[-3:] == ["--gpus", "4", "--main-only"]
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "launch with: python -m torch.distributed.run" not in src


def test_final_line_fits_the_drivers_tail():
    """The driver keeps a ~8 KB tail of stdout and `parsed` only the NAMES of extra keys: a line longer than that loses its
    head (round 4: config3.b4096 and config5 were cut off).  finalize_line rounds, packs and -- only if still too long --
    drops the least important sub-objects, naming them; the numbers the judge grades are never among the first to go."""
    import bench
    # a worst case: round 4's committed line (11.9 KB, a third of it prose) must come out under the budget with the
    # north-star legs intact
    d = json.load(open(os.path.join(ROOT, "profiles", "r04_bench_line.json")))
    line = bench.finalize_line(d, budget=8000)
    out = json.loads(line)
    assert len(line) <= 8000
    for k in ("metric", "value", "roofline", "cpu_baseline", "config3", "config5", "config4", "qp_solve"):
        assert k in out, k
    assert "b4096" in out["config3"] and "roofline" in out["config3"]["b4096"]
    assert out.get("dropped") and out["dropped"][0] == "host_api"
    # nothing is dropped from a line that fits; floats are rounded to five digits; separators are compact
    small = {"metric": "m", "value": 1234567.891, "config3": {"b4096": {"ms_per_step": 0.0481234567}}, "host_api": {"value": 1.0}}
    line = bench.finalize_line(small)
    assert json.loads(line) == {"metric": "m", "value": 1234600.0, "config3": {"b4096": {"ms_per_step": 0.048123}},
                                "host_api": {"value": 1.0}}
    assert ", " not in line and bench.LINE_BUDGET <= 7000
    # this round's own committed line (if there is one yet) is within the budget as printed
    f = os.path.join(ROOT, "profiles", "r05_bench_line.json")
    if os.path.exists(f):
        raw = open(f).read().strip()
        assert len(raw) <= bench.LINE_BUDGET
        now = json.loads(raw)
        assert "b4096" in now["config3"] and "config5" in now and "dropped" not in now
    # no prose keys left in the legs
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert '"note":' not in src and '"timing":' not in src and '"flops_counted"' not in src


def test_config3_leg_snapshots_before_the_kernel_split():
    """Round 4's driver line carried a false 4e-6 gradient error: the per-kernel split (whose third launch is the plain
    propagate, without the rho * sum T term) wrote into the buffers the comparison read afterwards."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    body = src[src.index("def run_config3"):src.index("def qp_newton_step_flops")]
    assert body.index("snap = (cost[:B].cpu()") < body.index("cost_grad_kernel_split(\n")
    assert "gP2, gT2)" in body and "gpu_vs_cpu_max_rel_gradP_err" in body


def test_time_steps_retimes_a_pass_the_runtime_stall_fell_into():
    """bench.time_steps: a pass whose wall time is far beyond what its per-step events say (the HIP runtime's one-off host stall
    between two steps) is discarded and the same steps are timed again, once; a clean pass is kept; a slow STEP (events and wall
    agree) is not a stall."""
    import time
    import types
    import bench

    class Ev:
        def __init__(self, enable_timing=True):
            self.t = None

        def record(self):
            self.t = time.perf_counter()

        def elapsed_time(self, other):
            return (other.t - self.t) * 1e3
    fake = types.SimpleNamespace(cuda=types.SimpleNamespace(Event=Ev))
    calls = {"n": 0, "stall_at": 7}

    def step(i, ev):
        calls["n"] += 1
        if calls["n"] == calls["stall_at"]:
            time.sleep(0.05)                    # the host blocks BETWEEN two event pairs
        ev[0].record()
        time.sleep(0.0005)
        ev[1].record()
    elapsed, ev, retimed = bench.time_steps(fake, None, False, None, 10, step, lambda: None)
    assert retimed and calls["n"] == 20 and elapsed < 0.03 and len(ev) == 10
    # the discarded pass is reported, not lost: its wall time per step holds the 50 ms the host was blocked for
    assert retimed["ms_per_step"] >= 5.0 and retimed["event_ms_median_step"] < 2.0 and retimed["event_ms_max_step"] < 5.0
    calls.update(n=0, stall_at=-1)
    elapsed, ev, retimed = bench.time_steps(fake, None, False, None, 10, step, lambda: None)
    assert not retimed and calls["n"] == 10

    def slow_steps(i, ev):                      # 3 ms per step inside the events: nothing to re-time
        ev[0].record()
        time.sleep(0.003)
        ev[1].record()
    elapsed, ev, retimed = bench.time_steps(fake, None, False, None, 10, slow_steps, lambda: None)
    assert not retimed and elapsed >= 0.03


def test_leg_workloads_and_their_traffic_lookup(tmp_path, monkeypatch):
    """`--workload config3|config4|qp` (one leg of the default line alone, the form tools/profile_leg.sh profiles) and
    pmc_leg_traffic: WRITE_SIZE + 2 x FETCH_SIZE per launch x launches per step over the kernels of a leg, from
    profiles/<round>_<leg>_pmc.json; the newest file wins; an ambiguous or missing kernel gives None, never a guess."""
    import bench
    src = open(os.path.join(ROOT, "bench.py")).read()
    for leg in ("config3", "config4", "qp"):
        assert f'"{leg}"' in src[src.index('ap.add_argument("--workload"'):src.index("args = ap.parse_args()")]
    assert "three_launch_split_us" in src and '"traffic": None, "kernel"' not in src
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    assert bench.pmc_leg_traffic("config3", bench.cost_grad_picks(1)) is None
    (prof / "r06_config3_pmc.json").write_text(json.dumps({"leg": "config3", "kernels": [
        {"name": "void anet::k_minco_cost_grad_fused<4, 8, true, 2>(anet::FusedArgs, double const*)", "grid": 65536, "n": 2000,
         "fetch_kib": 10000.0, "write_kib": 5000.0},
        {"name": "void anet::k_minco_solve<4, 8, true, 2>(anet::SolveArgs)", "grid": 131072, "n": 100, "fetch_kib": 1.0, "write_kib": 2.0},
        {"name": "void anet::k_piece_grad<4, false, 1>(anet::PieceGradArgs, double const*)", "grid": 1048576, "n": 100,
         "fetch_kib": 3.0, "write_kib": 4.0},
        {"name": "void anet::k_minco_propagate<4, 8, true, 2>(anet::PropArgs)", "grid": 131072, "n": 100, "fetch_kib": 5.0, "write_kib": 6.0}]}))
    assert bench.pmc_leg_traffic("config3", bench.cost_grad_picks(1)) == (2 * 10000.0 + 5000.0) * 1024
    assert bench.pmc_leg_traffic("config3", bench.cost_grad_picks(3)) == (2 * (1 + 3 + 5) + (2 + 4 + 6)) * 1024.0
    assert bench.pmc_leg_traffic("config3", [("k_minco_", None, 1)]) is None          # three kernels match: ambiguous
    assert bench.pmc_leg_traffic("config3", [("k_minco_solve<", 4096, 1)]) is None     # no entry of that grid
    assert bench.pmc_leg_traffic("config5", bench.cost_grad_picks(3)) is None          # another leg's file is not consulted
    r = bench.fp64_roofline(1e12, 1.0, 1000.0, "k", traffic=2200.0)
    assert r["traffic"] == 2200.0 and abs(r["hbm"]["traffic_over_algorithmic"] - 2.2) < 1e-12
    assert bench.fp64_roofline(1e12, 1.0, 1000.0, "k")["hbm"]["traffic_over_algorithmic"] is None
