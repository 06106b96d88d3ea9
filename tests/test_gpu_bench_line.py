"""The driver's command, `python bench.py --gpus 1 --steps 20 --warmup 5`, end to end on the GPU: ONE JSON line on stdout with the fields the
contract names (metric / value / roofline / cpu_baseline ...) and the side figures this build adds (hrotate, the generic chain, the one-launch
transform's placement counter)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_contract():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines            # one line, nothing else on stdout
    d = json.loads(lines[0])
    assert d["metric"] == "hmult+key-switch ops/sec" and d["unit"] == "ops/s" and d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5
    assert d["higher_is_better"] is True and d["vs_baseline"] is None and d["dtype"] == "u64" and d["data"] == "synthetic"
    assert d["value"] > 1000 and abs(d["value"] * d["ms_per_step"] * 1e-3 - 1.0) < 1e-6
    cfg = d["config"]
    assert "config_4.cfg hmult L=45 l=35 alpha=15" in cfg["workload"] and "model" not in cfg
    forced_generic = os.environ.get("HOMULATOR_ARITH") == "generic"   # (the suite also runs with every context forced onto the generic back-end)
    assert cfg["batch"] * cfg["streams"] == 20 and cfg["moduli"].startswith("generic" if forced_generic else "mont32")
    assert set(os.path.basename(p) for p in d["hip_library"]) >= {"libhomulator_hip.so", "libhm_gen.so" if forced_generic else "libhm_m32.so"}
    r = d["roofline"]
    assert r["contract_bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.05 < r["frac"] < 1.0
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["us_per_launch"] * 1e3)) < 1e-6 * r["achieved"]
    assert r["algorithmic_bytes_per_launch"] == 50 * 1048576 and r["traffic"] and r["traffic"] > r["algorithmic_bytes_per_launch"]
    ip = r["in_place"]      # the same sweep with in == out, beside the contract's figure
    assert 0.05 < ip["frac"] < 1.0 and abs(ip["achieved"] - r["algorithmic_bytes_per_launch"] / (ip["us_per_launch"] * 1e3)) < 1e-6 * ip["achieved"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "ops/s" and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    assert d["value"] / c["value"] > 50          # (a reported baseline, not the target)
    assert d["ntt_cross_xcd"] == {"after_sweep": 0, "after_timed_region": 0, "ntt_fused_small": 96}
    assert d["hrotate"]["ops_per_s"] > 1000 and d["hrotate"]["steps"] == 20
    assert d["hrotate"]["launches_per_op"] == 5 and d["hrotate"]["automorphism_launches"] == 0   # (pass 12: its readers gather through the automorphism)
    if not forced_generic:   # (the leg exists beside a headline on the default back-end)
        assert d["generic_chain_ops_per_s"] > 1000 and 0 < d["generic_chain_frac_evk_once"] < 1 and d["generic_chain"]["moduli"].startswith("generic")
    assert d["single_stream_ops_per_s"] > 1000 and d["sustained_ops_per_s"] > 1000
    assert d["measured_hbm"] and d["measured_hbm"]["batch"] == cfg["batch"] and d["measured_hbm"]["instances"] == cfg["streams"]
    assert d["roofline_op"] and d["roofline_op"]["frac"] > 0.1
    assert len(d["stage_us"]) == cfg["launches_per_op"] == 6
    # round 6: the op-level fractions inside `roofline`, the spread of `value` over three regions, medians for the side legs, the port's note
    assert abs(r["op_frac_evk_once"] - d["hmult_frac_evk_once"]) < 1e-12 and 0.3 < r["op_frac_evk_once"] < 1.0
    assert r["op_measured_frac"] is None or abs(r["op_measured_frac"] - d["measured_hbm"]["frac_of_peak"]) < 1e-12
    lo, med, hi = d["value_min_median_max"]
    assert lo <= med <= hi and lo <= d["value"] <= hi
    for leg in (d["hrotate"],) if forced_generic else (d["hrotate"], d["generic_chain"]):
        a, b_, c_ = leg["ops_per_s_min_median_max"]
        assert a <= b_ <= c_ and leg["ops_per_s"] == b_ and leg["regions"] == 3
    assert "__int128" in c["note"]
    for o_ in ("hadd", "pmult", "padd"):   # the reference's single-stage ops: timed in the driver's line too (streaming kernels: a good part of the HBM peak)
        e = d["elementwise"][o_]
        assert e["ops_per_s"] > 10000 and 0.2 < e["frac_of_hbm_peak"] < 1.0 and e["launches_per_op"] >= 1


@pytest.mark.parametrize("n", [2, 4])
def test_plain_multi_gpu_command_launches_its_own_ranks(n):
    """`python3 bench.py --gpus 2 --steps 8 --warmup 2` started PLAINLY (no torch.distributed.run, WORLD_SIZE unset): bench.py spawns the two
    ranks itself (fresh child processes; the parent never touches the GPU), relays rank 0's one JSON line and returns 0.  Rehearsed on the one
    GPU over gloo (HOMULATOR_DIST_BACKEND=gloo puts both ranks on device 0; RCCL refuses two ranks per device).  The line carries the sharded
    NTT sweep per rank and what the HIP library's communicator reports."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HOMULATOR_DIST_BACKEND"] = "gloo"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "8", "--warmup", "2"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == n and d["steps"] == 8 and d["value"] > 0 and d["config"]["transport"] == "gloo-rehearsal"
    sh = d["roofline"]["sharded"]
    own = [len([e for e in range(50) if e % n == r]) for r in range(n)]   # limb e of the extended basis -> rank e % n
    assert sh["limbs_per_rank"] == own and len(sh["us_per_rank"]) == n and sh["aggregate_gbs"] > 0 and all(0 < f < 1 for f in sh["frac_per_gpu"])
    assert d["roofline"]["algorithmic_bytes_per_launch"] == own[0] * 1048576
    assert d["comm"] == {"world": n, "ranks_seen": n, "transport": "external"}
    if n > 2:
        return
    # a rank that fails takes the job down with a non-zero exit code instead of leaving its peer in a collective
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "8", "--warmup", "2", "--op", "nonsense"], env=env, capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0
