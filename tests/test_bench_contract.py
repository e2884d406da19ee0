"""GPU: bench.py's JSON contract on a small domain (the driver runs bench.py at round end: a broken line is a lost measurement)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
            "config", "roofline", "cpu_baseline"]


def _bench(argv, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]  # ONE JSON line on stdout, nothing else
    return json.loads(lines[0])


@pytest.mark.parametrize("extra", [[], ["--batch", "1"], ["--batch", "1", "--in-flight", "1"], ["--workload", "commit"], ["--no-twiddle-cache"]],
                         ids=["default", "single_in_flight2", "one_at_a_time", "commit", "uncached"])
def test_bench_line_small_domain(extra):
    d = _bench(["--log-domain", "16", "--cpu-sample-log", "16", "--steps", "9", "--warmup", "2"] + extra)
    for k in REQUIRED:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 9 and d["warmup"] == 2 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["dtype"] == "u32" and d["data"] == "synthetic"
    assert d["value"] > 0 and abs(d["value"] - 4.0 * (1 << 16) * 9 / (d["ms_per_step"] * 9e-3)) / d["value"] < 1e-6
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and rf["unit"] == "GB/s" and 0 < rf["frac"] < 1 and "traffic" in rf and rf["traffic_source"]
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and cb["root_equals_gpu_root"] is True
    assert cb["multi_core"]["cores"] >= 1 and cb["multi_core"]["value"] > 0 and cb["multi_core"]["usable_cores"] >= cb["multi_core"]["cores"]
    assert cb["multi_core_all"]["all_usable_cores"] is True and cb["multi_core_all"]["cores"] == cb["multi_core_all"]["usable_cores"]  # every usable host core
    assert len(cb["reference_bench_sizes"]) == 5 and all(r["roots_equal"] and r["proofs_equal"] for r in cb["reference_bench_sizes"])
    if "commit" not in extra:
        assert d["verified_proofs"] == 9
        assert "commit_and_generate_proof" in d["config"]["workload"]
        e2e = d["end_to_end"]  # host blobs through frieda_prove_many / frieda_commit_many, pageable and page-locked, + a lone call
        for k in ("pageable_ms_per_blob", "pinned_ms_per_blob", "lone_call_ms", "pageable_commit_ms_per_blob", "pinned_commit_ms_per_blob", "device_resident_ms_per_blob"):
            assert e2e[k] > 0, k
    rec = d["reconstruct"]  # the sampling side: the blob back from one 1/16 block and from 2^L + 2 single points, checked inside bench.py
    assert rec["from_block_ms"] > 0 and rec["from_points_ms"] > 0 and rec["points"] == (1 << 12) + 2
    assert d["by_config"] == []  # nothing at or below 2^16 in BASELINE.json's configurations


def test_bench_line_carries_the_other_baseline_configs():
    """by_config: the 2^20 proof stream and the 2^22 commit stream beside a 2^22 headline (the driver's default run carries 2^20, 2^22 and
    the 2^24 commit beside the 2^24 headline)."""
    d = _bench(["--log-domain", "22", "--cpu-sample-log", "16", "--steps", "8", "--warmup", "1", "--batch-extra", "0", "--sequential-extra", "4"])
    cfg2 = [r for r in d["by_config"] if r["workload"].startswith("configs[1]")]
    assert len(cfg2) == 1 and cfg2[0]["log_domain"] == 20 and cfg2[0]["ms_per_pass"] > 0 and 0 < cfg2[0]["frac_of_hbm_peak_wall"] < 1
    rows = {(r["log_domain"], r["workload"]): r for r in d["by_config"] if not r["workload"].startswith("configs[1]")}
    assert set(rows) == {(20, "commit_and_generate_proof"), (22, "commit")}
    for r in rows.values():
        assert r["ms_per_blob"] > 0 and 0 < r["frac_of_hbm_peak_wall"] < 1 and r["lone_call"]["ms"] >= r["ms_per_blob"] * 0.5 and r["dominant_kernel"]["frac"] > 0
    assert rows[(20, "commit_and_generate_proof")]["verified_proofs"] == 512  # (a stream long enough for the policy's cut at this size, round 6)
    assert 0 < rows[(22, "commit")]["two_contexts"]["ms_per_blob"]  # the commit stream over two contexts taking turns
    r20 = rows[(20, "commit_and_generate_proof")]  # the library's batch policy cuts the 512 blobs into 2 calls of 256 (workspace bytes in flight: 43 GB per call)
    assert r20["measured_loop"].startswith("library batch policy (frieda_batch_plan): 2 calls of 256 / 256 blobs")
    fb = r20["fixed_batch4"]  # the 4-per-call cut of rounds 1-4, kept beside it for continuity
    assert fb["measured_loop"].startswith("4 blobs") and 0 < r20["ms_per_blob"] < fb["ms_per_blob"] * 1.05 and fb["verified_proofs"] == 512
    # round 6: what the fraction is, the counter traffic beside it, every call's latency, the environment defaults applied
    assert "byte model" in d["roofline"]["frac_is"] and d["roofline"]["top_family_by_summed_time"]["ms_per_step"] > 0
    pm = d["roofline_valu"][0]["vs_pipe_model"]  # the dominant launch against a pipe model that owes nothing to the product code
    assert 0 < pm["frac_at_2.4_ghz"] < pm["frac_at_probe_clock"] < 1 and pm["cycles_per_wave_compression"] == {"leaf": 1400, "node": 1904}
    assert d["call_latency"]["max_ms"] > 0 and sum(c["blobs"] for c in d["call_latency"]["calls"]) == 8
    assert set(d["env_defaults"]) == {"HSA_ENABLE_IPC_MODE_LEGACY", "NCCL_SOCKET_IFNAME"}
    v4 = d["value_fixed_batch4"]  # the headline's own workload through the 4-per-call cut of rounds 1 - 4 (like for like with BENCH_r01 .. r04)
    assert v4["measured_loop"].startswith("4 blobs per call") and v4["verified_proofs"] == 20 and v4["ms_per_blob"] > 0
    assert d["roofline"]["traffic"] is None or d["roofline"]["traffic"] > 0


def test_bench_forced_collective_one_rank():
    """RCCL init, root all_gather, barrier and max-reduce with a single rank: the N > 1 code path on a one-GPU box.  The line of that
    path is self-sufficient: cpu_baseline (computed by rank 0 after the process group is gone), roofline, roofline_valu, and the
    single-process frieda_prove_many leg from a fresh child process with its roots checked against the per-rank run's."""
    d = _bench(["--gpus", "1", "--log-domain", "16", "--cpu-sample-log", "16", "--steps", "6", "--warmup", "1", "--batch-extra", "0", "--sequential-extra", "0"],
               env_extra={"FRIEDA_BENCH_FORCE_DIST": "1"})
    assert d["n_gpus"] == 1 and d["verified_proofs"] == 6
    assert d["cpu_baseline"]["root_equals_gpu_root"] is True and d["cpu_baseline"]["multi_core_all"]["all_usable_cores"] is True
    assert d["roofline"]["frac"] > 0 and isinstance(d["roofline_valu"], list) and d["roofline_valu"][0]["frac"] > 0
    spm = d["single_process_multi"]
    assert spm.get("error") is None, spm
    hd = spm["headline"]
    assert spm["devices"] == [0] and hd["blobs"] == 8 and hd["verified_proofs"] == 8 and hd["roots_equal_per_rank_run"] == 6  # (the per-rank run knows 6 of the 8 seeds)
    assert hd["prove_ms_per_blob"] > 0 and hd["commit_ms_per_blob"] > 0 and hd["value"] > 0


@pytest.mark.parametrize("slots,log_domain", [(2, 16), (3, 16), (2, 22)])
def test_bench_single_process_leg_over_several_device_slots(slots, log_domain):
    """`bench.py --spm-child` (what an N > 1 line's rank 0 spawns) over 2 and 3 device slots of this one GPU against the RCCL test double:
    blob i -> slot i mod N, one root gather per call; at 2^22 also BASELINE configs[3] as written (seeds 100 .. 107) against the oracle's
    roots in tests/golden/config4_roots.json."""
    d = _bench(["--spm-child", "--gpus", str(slots), "--spm-devices", ",".join(["0"] * slots), "--log-domain", str(log_domain), "--spm-blobs-per-gpu", "2"],
               env_extra={"FRIEDA_RCCL_PATH": os.path.join(ROOT, "tests", "cpp", "librccl_stub.so")})
    assert d["devices"] == [0] * slots and d["uses_rccl"] is True and d["gather_count"] > 0
    hd = d["headline"]
    assert hd["blobs"] == 2 * slots and hd["verified_proofs"] == 2 * slots and len(set(hd["roots"])) == 2 * slots
    if log_domain >= 22:
        c4 = d["config4"]
        assert c4["blobs"] == 8 and c4["verified_proofs"] == 8 and c4["roots_equal_oracle_fixture"] is True
    else:
        assert "config4" not in d
