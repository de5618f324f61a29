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
