"""CPU: the N > 1 path — blobs sharded one per rank, all_gather of the 32-byte roots — with world_size 2 over gloo.
The commit function is injected (here: the oracle, as the checker) so the sharding and gather logic runs without a GPU."""
import os
import socket
import sys

import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_blobs, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    from conftest import splitmix64_bytes
    from frieda_amd import batch
    from oracle import oracle as O

    dist.init_process_group("gloo", rank=rank, world_size=world)
    blobs = [splitmix64_bytes(100 + i, 1000 + 37 * i).tobytes() for i in range(n_blobs)]
    calls = []

    def commit_fn(b):
        calls.append(len(b))
        return O.commit(b, 4)

    roots = batch.commit_batch(blobs, 4, commit_fn=commit_fn)
    q.put((rank, [r.hex() for r in roots], calls))
    dist.destroy_process_group()


@pytest.mark.parametrize("n_blobs", [2, 5])
def test_commit_batch_world2_gloo(oracle, n_blobs):
    from conftest import splitmix64_bytes

    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_blobs, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expected = [oracle.commit(splitmix64_bytes(100 + i, 1000 + 37 * i).tobytes(), 4).hex() for i in range(n_blobs)]
    for rank, roots, calls in results:
        assert roots == expected  # every rank ends with every root, in blob order
        assert len(calls) == len(range(rank, n_blobs, world))  # and hashed only its own shard


def _prove_worker(rank, world, port, n_blobs, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    from conftest import splitmix64_bytes
    from frieda_amd import batch
    from oracle import oracle as O

    dist.init_process_group("gloo", rank=rank, world_size=world)
    blobs = [splitmix64_bytes(200 + i, 700 + 11 * i).tobytes() for i in range(n_blobs)]
    seeds = [None if i % 2 else 50 + i for i in range(n_blobs)]
    cfg = O.make_config(6, 4, 0, 8)

    class Cfg:  # the shape batch.prove_batch reads the blow-up factor from
        class fri_config:
            log_blowup_factor = 4

    roots, proofs = batch.prove_batch(blobs, seeds, Cfg, prove_fn=lambda b, s: O.commit_and_generate_proof(b, s, cfg))
    ok = all(O.verify(p, seeds[i]) and bytes(p.c.first_layer.commitment) == roots[i] for i, p in proofs.items())
    q.put((rank, [r.hex() for r in roots], sorted(proofs), ok))
    dist.destroy_process_group()


def test_prove_batch_world2_gloo(oracle):
    from conftest import splitmix64_bytes

    world, port, n_blobs = 2, _free_port(), 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_prove_worker, args=(r, world, port, n_blobs, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expected = [oracle.commit(splitmix64_bytes(200 + i, 700 + 11 * i).tobytes(), 4).hex() for i in range(n_blobs)]
    for rank, roots, mine, ok in results:
        assert roots == expected and ok
        assert mine == list(range(rank, n_blobs, world))


def test_shard_indices():
    from frieda_amd.batch import shard_indices

    assert shard_indices(8, 3, 8) == [3]
    assert shard_indices(5, 0, 2) == [0, 2, 4]
    assert shard_indices(5, 1, 2) == [1, 3]
    assert sorted(sum((shard_indices(11, r, 4) for r in range(4)), [])) == list(range(11))


def _gather_worker(rank, world, port, k, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    from frieda_amd import batch

    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = b"".join(bytes([(7 * rank + i) % 256]) * 32 for i in range(k))  # K distinguishable 32-byte roots
    g = batch.gather_rank_roots(mine)
    q.put((rank, [bytes(g[r].numpy()) for r in range(world)]))
    dist.destroy_process_group()


def test_bench_root_exchange_world2_gloo():
    """bench.py's only collective: after the K timed steps every rank contributes its K roots to one all_gather."""
    world, port, k = 2, _free_port(), 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gather_worker, args=(r, world, port, k, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, rows in results:
        for r in range(world):
            assert rows[r] == b"".join(bytes([(7 * r + i) % 256]) * 32 for i in range(k)), (rank, r)



def _run_bench(argv, env_extra=None, timeout=300):
    import json
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, capture_output=True, text=True, timeout=timeout, env=env)
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    return r, [json.loads(ln) for ln in lines if ln.startswith("{")]


@pytest.mark.parametrize("gpus", [2, 3, 8])
def test_bench_self_launches_its_ranks(gpus):
    """`python bench.py --gpus N` (no torch.distributed.run, no WORLD_SIZE) spawns its N ranks itself and relays rank 0's single
    JSON line.  `--dry-collective gloo` keeps the launcher, rendezvous, barriers, root all_gather and max-reduce and stubs the GPU."""
    r, docs = _run_bench(["--gpus", str(gpus), "--steps", "5", "--warmup", "1", "--dry-collective", "gloo", "--cpu-sample-log", "12"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(docs) == 1 and r.stdout.count("\n") == 1
    d = docs[0]
    assert d["n_gpus"] == gpus and d["steps"] == 5 and d["dry_run"] is True and d["scaling"] == "weak"
    assert d["roots_gathered"] == gpus * 5
    # an N > 1 line is self-sufficient (north_star: the CPU figure "in the same run" at 1, 2, 4 and 8 GPUs): rank 0 adds the CPU legs
    # after the process group is gone, and the single-process frieda_prove_many leg comes from a fresh child process
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0
    assert cb["multi_core"]["value"] > 0 and cb["multi_core_all"]["all_usable_cores"] is True and cb["multi_core_all"]["cores"] == len(os.sched_getaffinity(0))
    spm = d["single_process_multi"]
    assert spm.get("error") is None and spm["dry_run"] is True and spm["devices"] == list(range(gpus))
    assert "roofline" in d and "roofline_valu" in d


def test_bench_single_process_leg_failure_is_reported_in_place():
    """The extra leg must never cost the line its headline: a child that cannot start its devices is an `error` entry."""
    r, docs = _run_bench(["--gpus", "2", "--steps", "2", "--dry-collective", "gloo", "--no-cpu-baseline", "--spm-timeout", "0.001"])
    assert r.returncode == 0 and len(docs) == 1
    assert "error" in docs[0]["single_process_multi"] and docs[0]["n_gpus"] == 2 and docs[0]["cpu_baseline"] is None


def test_bench_under_an_external_launcher_does_not_respawn():
    """Started the way the driver starts it (torch.distributed.run sets WORLD_SIZE / RANK), bench.py must not spawn anything."""
    import subprocess

    port = _free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
           str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-collective", "gloo", "--cpu-sample-log", "10"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and '"n_gpus": 2' in lines[0]
    import json

    d = json.loads(lines[0])  # under the driver's launcher the line carries the same blocks
    assert d["cpu_baseline"]["value"] > 0 and d["single_process_multi"]["dry_run"] is True


def test_bench_self_launch_reports_a_failing_rank():
    """A rank that dies takes the launch down with a non-zero status instead of leaving the others in the rendezvous."""
    r, docs = _run_bench(["--gpus", "2", "--steps", "2", "--dry-collective", "gloo"], env_extra={"FRIEDA_BENCH_TEST_FAIL_RANK": "1"}, timeout=120)
    assert r.returncode != 0 and docs == []


def test_bench_refuses_more_ranks_than_visible_gpus():
    """`python bench.py --gpus N` with fewer than N visible devices must say so and exit non-zero before starting any rank (on a
    box with 8 GPUs the driver's `--gpus 8` then just works; on a box without, nobody waits in a rendezvous for missing ranks)."""
    import torch

    have = torch.cuda.device_count()
    r, docs = _run_bench(["--gpus", str(have + 2), "--steps", "2"], timeout=120)
    assert r.returncode == 2 and not docs
    assert f"--gpus {have + 2} but only {have} GPU(s) are visible" in r.stderr
