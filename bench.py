#!/usr/bin/env python3
"""bench.py — FRI-DAS commit/prove throughput on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one synthetic blob per GPU: `commit_and_generate_proof`
(/root/reference/src/proof.rs:32-77, the benches/proof.rs:30-44 workload) on a device-resident blob whose felts
exactly fill a 2^n domain at log_blowup_factor 4 (default n = 24 = BASELINE.json configs[4], the largest single-GPU
configuration): unpack -> 4 x circle NTT -> first Merkle tree -> every FRI fold + per-layer Merkle tree -> last-layer
interpolation -> grind (pow_bits 20) -> 20 query openings.  value = M31 field elements committed per second =
n_gpus * 4 * 2^n * steps / wall time, inputs already resident in HBM when the timed region starts.

The K timed steps process K DISTINCT blobs per GPU (a stream of blobs, as a data-availability node sees them) through the batched
entry points (every kernel is launched once per call, so the Fiat-Shamir latency chain is paid once per call), cut into calls by
the library's own batch policy (`frieda_batch_plan`, include/frieda_hip.h: workspace bytes in flight — up to 16 blobs per call at the 2^24
domain, one call per context at 2^22 and below; the number of calls a multiple of `--in-flight`, sizes equal to within one; `--batch B` forces about
B per call instead, as rounds 1-4 did with B = 4) with `--in-flight` calls in flight (default 2: one context = stream +
workspace each, so that chain also runs under the chip-filling kernels of the other call).  Results are those of K separate
calls (tests/test_gpu_parity.py).  Every one of the K timed proofs is verified after the timed region and the K roots must be
distinct; `sequential` in the JSON line is the one-proof-at-a-time figure (`--batch 1 --in-flight 1` makes it the headline;
`--batch 1` is two single proofs in flight).

Multi-GPU: one process per GPU (torch.distributed, backend nccl = RCCL), independent blobs sharded one per rank (weak
scaling); nothing is exchanged while blobs are processed — every rank keeps the roots of its K blobs and the ranks exchange
them in ONE all_gather of K x 32 bytes per rank after the last step, inside the timed region.  `python bench.py --gpus N`
launches its own N ranks when it was not started by torch.distributed.run (no WORLD_SIZE in the environment): the parent
never touches the GPU, spawns one child per rank (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT set),
relays rank 0's JSON line and exits with the worst child status.  `--dry-collective gloo` runs the same launcher, rendezvous,
barriers, root all_gather and max-reduce on CPU tensors with the GPU work stubbed out (the CPU test of the N > 1 plumbing;
its line carries "dry_run": true and value 0).

The JSON line also carries
  roofline     — the dominant launch (dominant_launch(): longest average launch among the kernel families within a factor two of the
                 largest summed HIP-event time on the kernels' own stream, instrumented replay of
                 the same K steps): achieved = algorithmic bytes / time against the 8 TB/s HBM peak;
  cpu_baseline — the CPU oracle (oracle/, a restated port of the reference's single-threaded CPU path) timed on this
                 host on a bounded sample (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def dist_env_defaults(env):
    """Environment defaults for RCCL on ONE node, applied with setdefault (whatever the launcher exported wins) and reported in the JSON
    line (`env_defaults`) so a reader sees what this process added.
    * HSA_ENABLE_IPC_MODE_LEGACY=0: the host driver of this pool only supports dmabuf IPC; without it RCCL fails with
      `hipIpcGetMemHandle: invalid argument` (the pool exports it itself; this only covers a shell that dropped it).
    * NCCL_SOCKET_IFNAME=lo: ONLY when the rendezvous itself is on the loopback (MASTER_ADDR unset / 127.0.0.1 / localhost / ::1), i.e.
      every rank is on this node and RCCL's bootstrap needs no NIC — left to itself it picks the container's first interface, whose name
      may not resolve (a one-rank communicator has been seen to take minutes to come up on such a box).  With any other MASTER_ADDR
      (a multi-node launch) RCCL keeps its own choice."""
    applied = {}
    for key, val, cond in (("HSA_ENABLE_IPC_MODE_LEGACY", "0", True),
                           ("NCCL_SOCKET_IFNAME", "lo", env.get("MASTER_ADDR", "127.0.0.1") in ("127.0.0.1", "localhost", "::1"))):
        if key in env:
            applied[key] = "inherited: " + env[key]
        elif cond:
            env[key] = val
            applied[key] = "set by bench.py: " + val
        else:
            applied[key] = "left to RCCL (rendezvous is not on the loopback)"
    return applied


def splitmix64_bytes(seed, n):
    cnt = (n + 7) // 8
    z = (np.uint64(seed) + np.uint64(0x9E3779B97F4A7C15) * np.arange(1, cnt + 1, dtype=np.uint64)).astype(np.uint64)
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    z = z ^ (z >> np.uint64(31))
    return z.view(np.uint8)[:n].copy()


def blob_len_for(log_domain, log_blowup=4):
    return (4 << (log_domain - log_blowup)) * 30 // 8


def algorithmic_bytes(n, workload, log_blowup=4, log_last=0):
    """SURVEY.md §8d byte model: every logical stage reads its input once and writes its output once."""
    N = float(1 << n)
    enc = 16.0 * N * (1.0 + 2.0 ** (-log_blowup))
    tree = lambda m: 144.0 * m  # noqa: E731
    total = enc + tree(N)
    if workload == "prove":
        total += 24.0 * N  # fold_circle_into_line
        m = N / 2
        last = float(1 << (log_last + log_blowup))
        while m > last:
            total += tree(m) + 24.0 * m  # per inner layer: tree + fold_line
            m /= 2
    return total


def dominant_launch(kern):
    """The kernel `roofline` is quoted on.  The timer's families are named by what a launch produces, and one family can hold launches of
    very different sizes and kernels (tree5_fold_line: thirteen launches from 2^22 leaves down, three kernel templates), so "the family
    with the largest summed time" can flip between two near-equal families from run to run and its average launch then describes no
    launch in particular.  Taken instead: among the families within a factor two of the largest summed time, the one with the longest
    AVERAGE launch — at the headline size the single 2^24-leaf launch of the first tree (also the largest kernel symbol of the rocprofv3
    summary under profiles/)."""
    top = max(k["total_ms"] for k in kern)
    cand = [k for k in kern if k["total_ms"] >= 0.5 * top and k["launches"] > 0]
    return max(cand, key=lambda k: k["total_ms"] / k["launches"]) if cand else kern[0]


def traffic_from_profiles(kernel, n, workload):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE collected
    in separate runs and corrected as MI355X_MICROARCH.md §HBM prescribes; tools/traffic_from_pmc.py).  Only valid for the
    configuration the counters were taken on (2^24 domain).  Returns (bytes or None, source description, per-blob bytes of the same
    kernel in the measured loop's batched mode or None): committed measurements of an earlier run of this same command, NOT something
    this run measured.  `traffic` is the lone-proof figure: the mode of the instrumented replay that `achieved` and `avg_launch_us`
    come from."""
    if n != 24:
        return None, f"none: the committed PMC passes were taken on the 2^24 domain, this run is 2^{n}", None
    import glob

    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_prove24_traffic.json")), reverse=True):  # newest round first
        name = os.path.basename(path)
        try:
            doc = json.load(open(path))
            val = doc["kernels"][kernel]["traffic_bytes_per_launch"]
        except (KeyError, ValueError):
            continue
        src = f"profiles/{name} (committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command in lone-proof mode, --batch 1 --in-flight 1"
        if doc.get("collected_at_commit"):
            src += f", collected at commit {doc['collected_at_commit']}"
        if doc.get("csrc_sha16"):
            src += ("; frieda_amd/csrc unchanged since" if doc["csrc_sha16"] == _csrc_sha16() else "; STALE: frieda_amd/csrc has changed since the counters were taken")
        bat = None
        try:
            b = doc["batched"]
            bat = b["kernels"][kernel]["traffic_bytes_per_launch"] / b["blobs_per_launch"]
        except (KeyError, TypeError):
            pass
        return val, src + "; not re-measured in this run)", bat
    return None, "none: no committed PMC pass names this kernel", None


def _csrc_sha16():
    """the same fingerprint tools/traffic_from_pmc.py stores in the traffic file"""
    import hashlib

    root = os.path.join(ROOT, "frieda_amd", "csrc")
    h = hashlib.sha256()
    for name in sorted(os.listdir(root)):
        if name.endswith((".hip", ".h", ".cpp")):
            h.update(name.encode())
            h.update(open(os.path.join(root, name), "rb").read())
    return h.hexdigest()[:16]


def _host_threads():
    """threads of the multi-core CPU baseline: the cores this process may run on (sched_getaffinity), capped at
    FRIEDA_BENCH_CPU_THREADS (default 32 = one GPU's share of a 256-core, 8-GPU host; each thread holds ~0.7 GB of oracle workspace
    at the 2^22 sample, so the cap also bounds host memory)"""
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cap = int(os.environ.get("FRIEDA_BENCH_CPU_THREADS", "32"))
    return max(1, min(avail, cap))


def cpu_baseline(sample_log, workload, calls, all_cores_log=22):
    """The CPU oracle (oracle/, a restated port of the reference's single-threaded CPU path; the reference itself is Rust +
    un-vendored stwo and cannot be built here) timed on this host: (1) one thread — the reference has no parallelism at all
    (stwo pulled without `parallel`, Cargo.toml:12); (2) `multi_core` / `multi_core_all`: one independent blob per core on
    min(usable cores, 32) and on EVERY usable core, one single-threaded worker process each (cpu_multi_child; the
    line says how many cores and whether that is every usable one)."""
    from oracle import oracle as O

    O.build()
    cfg = O.make_config(20, 4, 0, 20)
    name = "commit_and_generate_proof" if workload == "prove" else "commit"

    def run(data):
        if workload == "prove":
            return O.commit_and_generate_proof(data, data.size, cfg)[0]
        return O.commit(data, 4)

    data = splitmix64_bytes(100, blob_len_for(sample_log))
    t0 = time.perf_counter()
    root = None
    for _ in range(calls):
        root = run(data)
    dt = time.perf_counter() - t0
    out = {
        "root": bytes(root).hex(),
        "value": 4.0 * (1 << sample_log) * calls / dt,
        "unit": "M31 field-elems/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{calls} x {name} on a 2^{sample_log} domain (same generator and config as the GPU workload), {dt:.1f} s of "
        f"single-thread CPU on a {os.cpu_count()}-core host; restated CPU path (oracle/), not the upstream Rust binary",
    }
    # many cores: one blob per core, one PROCESS per core, from a child interpreter that never touches the GPU (the many-core legs fork
    # their workers; this process has initialised the GPU and must not be forked)
    import subprocess

    argv = [sys.executable, os.path.abspath(__file__), "--cpu-multi-child", "--workload", workload, "--cpu-sample-log", str(all_cores_log)]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run(argv, env=env, capture_output=True, text=True, timeout=900)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or len(lines) != 1:
        raise RuntimeError(f"--cpu-multi-child exited {r.returncode}: {r.stderr[-800:]}")
    out.update(json.loads(lines[0]))
    return out


def cpu_multi_child(args):
    """`bench.py --cpu-multi-child`: the many-core legs of the CPU baseline, one independent blob per core, one forked worker PROCESS per
    core (256 threads of one process serialise on its address-space lock while the oracle's buffers fault in: a 2^20 proof per thread
    took 25 s on the 256-core GPU host, slower in aggregate than 32 threads).  `multi_core`: min(usable cores, FRIEDA_BENCH_CPU_THREADS =
    32) workers, one proof of the sample domain each — one GPU's share of an 8-GPU host; `multi_core_all`: EVERY usable core, one 2^20
    proof each (~45 MB of oracle workspace per worker).  Prints one JSON line."""
    import multiprocessing as mp

    from oracle import oracle as O

    O.build()
    O.lib()  # loaded before the fork: the workers inherit it
    workload = args.workload
    name = "commit_and_generate_proof" if workload == "prove" else "commit"
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1

    def leg(workers, log, what):
        ctx = mp.get_context("fork")
        start, q = ctx.Barrier(workers + 1), ctx.Queue()

        def work(i):
            from oracle import oracle as O2

            data = splitmix64_bytes(100 + i, blob_len_for(log))
            cfg = O2.make_config(20, 4, 0, 20)
            start.wait()
            root = O2.commit_and_generate_proof(data, data.size, cfg)[0] if workload == "prove" else O2.commit(data, 4)
            q.put((i, bytes(root).hex()))

        procs = [ctx.Process(target=work, args=(i,)) for i in range(workers)]
        for p_ in procs:
            p_.start()
        start.wait()  # every worker holds its blob and stands at the line
        t0 = time.perf_counter()
        roots = dict(q.get(timeout=800) for _ in range(workers))
        dta = time.perf_counter() - t0
        for p_ in procs:
            p_.join(timeout=60)
        return {
            "value": 4.0 * (1 << log) * workers / dta,
            "unit": "M31 field-elems/s",
            "cores": workers,
            "usable_cores": usable,
            "all_usable_cores": workers == usable,
            "nproc": os.cpu_count(),
            "sample": f"{workers} distinct blobs, one {name} on a 2^{log} domain per worker process, {workers} single-threaded workers = {what} this "
            f"process may run on ({usable} of {os.cpu_count()} on the host), {dta:.1f} s wall",
            "first_root": roots[0],
        }

    all_cores_log = args.cpu_sample_log
    workers = _host_threads()
    out = {"multi_core": leg(workers, all_cores_log, "every core" if workers == usable else "a capped share of the cores")}
    every_log = min(20, all_cores_log)
    out["multi_core_all"] = out["multi_core"] if (workers == usable and every_log == all_cores_log) else leg(usable, every_log, "every core")
    sys.stdout.write(json.dumps(out) + "\n")
    return 0


def reference_bench_sizes(ctx, frieda_amd, torch):
    """The reference's own bench inputs (benches/commit.rs:6-10: (i % 256) for 1024 / 4096 / 16384 / 65536 bytes + the blob
    fixture; benches/proof.rs:5-12 config, seed = Some(len)): CPU oracle (1 thread) beside the GPU path, roots compared."""
    from oracle import oracle as O

    O.build()
    cfg = frieda_amd.PcsConfig(frieda_amd.FriConfig(4, 0, 20), 20)
    ocfg = O.make_config(20, 4, 0, 20)
    inputs = [(f"pattern:{n}", (np.arange(n, dtype=np.uint64) % 256).astype(np.uint8)) for n in (1024, 4096, 16384, 65536)]
    blob_path = os.path.join(ROOT, "tests", "golden", "blob")
    if os.path.exists(blob_path):
        inputs.append(("blob", np.frombuffer(open(blob_path, "rb").read(), dtype=np.uint8)))
    rows = []
    for name, data in inputs:
        d = torch.from_numpy(data.copy()).cuda()
        d_root = torch.zeros(32, dtype=torch.uint8, device="cuda")

        def timed(fn, reps):
            fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                r = fn()
            ctx.synchronize()
            return (time.perf_counter() - t0) / reps * 1e3, r

        gpu_commit_ms, _ = timed(lambda: ctx.commit_device(d.data_ptr(), data.size, 4, d_root.data_ptr()), 50)
        gpu_root = bytes(d_root.cpu().numpy())
        gpu_prove_ms, (p_root, proof) = timed(lambda: ctx.commit_and_generate_proof_device(d.data_ptr(), data.size, data.size, cfg), 20)
        t0 = time.perf_counter()
        c_root = bytes(O.commit(data, 4))
        cpu_commit_ms = (time.perf_counter() - t0) * 1e3
        t0 = time.perf_counter()
        o_root, o_proof = O.commit_and_generate_proof(data, data.size, ocfg)
        cpu_prove_ms = (time.perf_counter() - t0) * 1e3
        rows.append(
            {
                "input": name,
                "bytes": int(data.size),
                "cpu_commit_ms": cpu_commit_ms,
                "gpu_commit_ms": gpu_commit_ms,
                "cpu_prove_ms": cpu_prove_ms,
                "gpu_prove_ms": gpu_prove_ms,
                "roots_equal": gpu_root == c_root == p_root == bytes(o_root),
                "proofs_equal": proof.serialize() == o_proof.serialize(),
            }
        )
    return rows


def measure_config(frieda_amd, torch, device, n, workload, K, BSZ, D, cfg):
    """The measured loop of main() on another BASELINE configuration (benches/commit.rs:6-13,
    benches/proof.rs:30-44): K distinct device-resident blobs of a 2^n domain (generator seeds 100 + i) through
    `commit_and_generate_proof` (BSZ per call — 0: the library's batch policy —, D calls in flight) or `commit`, every proof verified; then the lone-call latency and the
    dominant kernel of an instrumented replay.  Fractions are algorithmic bytes / time against the 8 TB/s HBM peak."""
    blob_len = blob_len_for(n)
    seed = blob_len
    blobs = torch.empty((K, blob_len), dtype=torch.uint8, device="cuda")
    for i in range(K):
        blobs[i].copy_(torch.from_numpy(splitmix64_bytes(100 + i, blob_len)))
    ctx = frieda_amd.Context(device)
    roots_dev = torch.zeros(32 * K, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    elems = 4.0 * (1 << n)
    path_bytes = algorithmic_bytes(n, workload)
    out = {"log_domain": n, "workload": "commit_and_generate_proof" if workload == "prove" else "commit", "blob_bytes": blob_len, "blobs": K}
    if workload == "prove":
        def stream(bsz):
            """bsz 0: the library's batch policy cuts the K blobs into calls (BatchPipeline.run_stream_device); else bsz per call"""
            pipe = frieda_amd.BatchPipeline(device, D) if bsz != 1 else frieda_amd.ProofPipeline(device, D)

            def run():
                if bsz == 0:
                    return pipe.run_stream_device(blobs[0].data_ptr(), blob_len, blob_len, K, [seed] * K, cfg)
                res = []
                for i in range(0, K, bsz):
                    cnt = min(bsz, K - i)
                    if bsz > 1:
                        r = pipe.submit_device(blobs[i].data_ptr(), blob_len, blob_len, cnt, [seed] * cnt, cfg)
                        if r is not None:
                            res.extend(r)
                    else:
                        r = pipe.submit_device(blobs[i].data_ptr(), blob_len, seed, cfg)
                        if r is not None:
                            res.append(r)
                res.extend(pipe.drain())
                return res

            run()  # sizes the workspaces, builds the twiddles
            run()
            torch.cuda.synchronize()
            import gc

            gc.disable()  # (no collector pause inside the timed region: see the measured loop of main)
            t0 = time.perf_counter()
            res = run()
            torch.cuda.synchronize()
            dt_ = (time.perf_counter() - t0) / K
            gc.enable()
            assert len(res) == K and len({r for r, _ in res}) == K
            for r, p in res:
                assert p.commitment == r and frieda_amd.verify(p, seed), "a timed proof does not verify"
            cut = pipe.plan(blob_len, K, cfg) if bsz == 0 else None
            pipe.close()
            return dt_, [r for r, _ in res], cut

        dt, roots, cut = stream(BSZ)
        out["measured_loop"] = (f"library batch policy (frieda_batch_plan): {len(cut)} calls of {max(cut)} / {min(cut)} blobs, {D} calls in flight" if BSZ == 0 else
                                f"{BSZ} blobs per call, {D} calls in flight")
        out["verified_proofs"] = K
        root0 = roots[0]
        if BSZ == 0:  # for continuity with rounds 1-4, whose loop handed 4 blobs to every call whatever their size
            dt4, roots4, _ = stream(4)
            assert roots4 == roots
            out["fixed_batch4"] = {"measured_loop": f"4 blobs per call, {D} calls in flight (the cut of rounds 1-4)", "ms_per_blob": 1e3 * dt4, "value": elems / dt4,
                                   "frac_of_hbm_peak_wall": path_bytes / dt4 / 1e9 / HBM_PEAK_GBS, "verified_proofs": K}

        def lone():
            return ctx.commit_and_generate_proof_device(blobs[0].data_ptr(), blob_len, seed, cfg)
    else:
        def run():
            for i in range(K):
                ctx.commit_device(blobs[i].data_ptr(), blob_len, 4, roots_dev.data_ptr() + 32 * i)
            ctx.synchronize()

        run()
        run()
        t0 = time.perf_counter()
        run()
        dt = (time.perf_counter() - t0) / K
        out["measured_loop"] = "commit_device per blob, asynchronous, one context"
        root0 = bytes(roots_dev[:32].cpu().numpy())
        all_roots = bytes(roots_dev.cpu().numpy())
        assert len({all_roots[32 * i : 32 * i + 32] for i in range(K)}) == K
        # the same stream over two contexts (streams) taking turns: one blob's narrow tree tops run under the other's wide launches
        ctx2 = frieda_amd.Context(device)

        def run2():
            for i in range(K):
                (ctx if i % 2 == 0 else ctx2).commit_device(blobs[i].data_ptr(), blob_len, 4, roots_dev.data_ptr() + 32 * i)
            ctx.synchronize()
            ctx2.synchronize()

        roots_dev.zero_()
        run2()
        run2()
        t0 = time.perf_counter()
        run2()
        dt2 = (time.perf_counter() - t0) / K
        assert bytes(roots_dev.cpu().numpy()) == all_roots
        ctx2.close()
        out["two_contexts"] = {"measured_loop": "commit_device per blob, asynchronous, two contexts taking turns", "ms_per_blob": 1e3 * dt2,
                               "value": elems / dt2, "frac_of_hbm_peak_wall": path_bytes / dt2 / 1e9 / HBM_PEAK_GBS}

        def lone():
            ctx.commit_device(blobs[0].data_ptr(), blob_len, 4, roots_dev.data_ptr())
            ctx.synchronize()
            return bytes(roots_dev[:32].cpu().numpy()), None

    out.update({"ms_per_blob": 1e3 * dt, "value": elems / dt, "unit": "M31 field-elems/s", "frac_of_hbm_peak_wall": path_bytes / dt / 1e9 / HBM_PEAK_GBS,
                "root": root0.hex()})
    # a lone call: one blob at a time, synchronised
    reps = 20
    lone()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        r_l, _p = lone()
    torch.cuda.synchronize()
    dtl = (time.perf_counter() - t0) / reps
    assert r_l == root0
    out["lone_call"] = {"ms": 1e3 * dtl, "frac_of_hbm_peak": path_bytes / dtl / 1e9 / HBM_PEAK_GBS}
    # dominant kernel of a lone call (HIP events on the context's stream)
    ctx.set_kernel_timing(True)
    for _ in range(5):
        lone()
    kern = ctx.kernel_timing_report(reset=True)
    ctx.set_kernel_timing(False)
    if kern:
        kern.sort(key=lambda k: -k["total_ms"])
        dom = dominant_launch(kern)
        ach = dom["alg_bytes"] / (dom["total_ms"] * 1e-3) / 1e9 if dom["total_ms"] > 0 else 0.0
        out["dominant_kernel"] = {"kernel": dom["name"], "avg_launch_us": 1e3 * dom["total_ms"] / max(dom["launches"], 1), "achieved_GBps": ach,
                                  "frac": ach / HBM_PEAK_GBS, "gpu_kernel_ms_per_call": sum(k["total_ms"] for k in kern) / 5,
                                  "launches_per_call": sum(k["launches"] for k in kern) / 5}
    ctx.close()
    del blobs
    return out


def measure_config2(frieda_amd, torch, device, n=20):
    """BASELINE.json configs[1] as written: a 2^n-element M31 NTT (4 columns, blow-up 16) + fold_circle_into_line
    + one fold_line through the trait-granular Level B entry points (frieda_circle_evaluate / frieda_fold_circle_into_line /
    frieda_fold_line: what a Rust HipBackend's PolyOps::evaluate and FriOps would call), device-resident, fixed alphas.  Bit-exactness of
    exactly this sequence against the oracle is tests/test_gpu_parity.py::test_config2_ntt_plus_fold_round."""
    import ctypes as C

    L = n - 4
    ctx = frieda_amd.Context(device)
    lib, h = ctx._L, ctx._h
    g = torch.Generator(device="cpu").manual_seed(2)
    coef = torch.randint(0, 2**31 - 1, (4, 1 << L), dtype=torch.int32, generator=g).cuda()
    ev = torch.empty((4, 1 << n), dtype=torch.int32, device="cuda")
    l1 = torch.zeros((4, 1 << (n - 1)), dtype=torch.int32, device="cuda")
    l2 = torch.empty((4, 1 << (n - 2)), dtype=torch.int32, device="cuda")
    alphas = (C.c_uint32 * 4)(11, 22, 33, 44), (C.c_uint32 * 4)(55, 66, 77, 88)

    def once_three():  # the three trait-granular calls: four launches
        rc = lib.frieda_circle_evaluate(h, coef.data_ptr(), 4, L, n, ev.data_ptr())
        rc |= lib.frieda_fold_circle_into_line(h, l1.data_ptr(), ev.data_ptr(), n, alphas[0])
        rc |= lib.frieda_fold_line(h, l1.data_ptr(), n - 1, n, alphas[1], l2.data_ptr())
        assert rc == 0

    def once():  # the same three results in one pass over the evaluation: the folds ride in the transform's last pass (two launches)
        assert lib.frieda_circle_evaluate_fold2(h, coef.data_ptr(), L, n, ev.data_ptr(), alphas[0], 0, l1.data_ptr(), alphas[1], l2.data_ptr()) == 0

    def timed(fn):
        for _ in range(5):
            fn()
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        ctx.synchronize()
        return (time.perf_counter() - t0) / reps

    reps = 200
    # both forms leave the same three buffers (line 1 starts from zero for the accumulating trait call)
    l1.zero_()
    torch.cuda.synchronize()  # (torch's stream, not the context's)
    once_three()
    ctx.synchronize()
    want = (ev.clone(), l1.clone(), l2.clone())
    torch.cuda.synchronize()
    once()
    ctx.synchronize()
    assert torch.equal(ev, want[0]) and torch.equal(l1, want[1]) and torch.equal(l2, want[2]), "frieda_circle_evaluate_fold2 differs from the three calls"
    dt3 = timed(once_three)
    dt = timed(once)
    N = float(1 << n)
    alg = 16.0 * N * (1.0 + 1.0 / 16.0) + 24.0 * N + 24.0 * N / 2  # encode + fold_circle_into_line + fold_line at N / 2
    ctx.close()
    return {"log_domain": n, "workload": "configs[1]: circle NTT (4 columns) + fold_circle_into_line + one fold_line, Level B entry points",
            "ms_per_pass": 1e3 * dt, "value": 4.0 * N / dt, "unit": "M31 field-elems/s", "algorithmic_bytes": alg,
            "frac_of_hbm_peak_wall": alg / dt / 1e9 / HBM_PEAK_GBS, "launches": "asynchronous on one context, 200 passes",
            "entry_point": "frieda_circle_evaluate_fold2 (the folds ride in the transform's last pass: two launches); results checked equal to the three calls",
            "three_calls": {"ms_per_pass": 1e3 * dt3, "frac_of_hbm_peak_wall": alg / dt3 / 1e9 / HBM_PEAK_GBS,
                            "entry_points": "frieda_circle_evaluate + frieda_fold_circle_into_line + frieda_fold_line (four launches)"}}


def reconstruct_side(frieda_amd, torch, device, n):
    """The other side of data-availability sampling (/root/reference/README.md:56-69), device-resident and checked:
    the blob of the headline size back from (i) one aligned 1/16 block of its codeword and (ii) 2^L + 2 single points sampled anywhere
    in it (erasure-locator route: no linear system; every sample compared with the re-encoded result).  Not part of `value`."""
    import numpy as np

    L = n - 4
    blob_len = blob_len_for(n)
    ctx = frieda_amd.Context(device)
    lib, h = ctx._L, ctx._h
    data = torch.from_numpy(splitmix64_bytes(100, blob_len)).cuda()
    coef = torch.empty((4, 1 << L), dtype=torch.int32, device="cuda")
    ev = torch.empty((4, 1 << n), dtype=torch.int32, device="cuda")
    assert lib.frieda_unpack30(h, data.data_ptr(), blob_len, coef.data_ptr(), 4 << L) == 0
    assert lib.frieda_circle_evaluate(h, coef.data_ptr(), 4, L, n, ev.data_ptr()) == 0
    out_bytes = torch.empty(blob_len + 8, dtype=torch.uint8, device="cuda")
    g = torch.Generator(device="cpu").manual_seed(3)
    n_pts = (1 << L) + 2
    pos = torch.randperm(1 << n, generator=g)[:n_pts]
    idx = np.ascontiguousarray(pos.numpy().astype(np.uint32))
    cells = ev[:, pos.cuda()].t().contiguous()  # [n_pts][4][1]: the layout of frieda_reconstruct_points_device with cells of one entry
    block = ev[:, 5 << L : 6 << L].contiguous()

    def best(fn, reps=3):
        t = None
        for _ in range(reps + 1):  # first call sizes the workspace
            out_bytes.zero_()
            ctx.synchronize()
            t0 = time.perf_counter()
            assert fn() == 0
            ctx.synchronize()
            d = time.perf_counter() - t0
            assert torch.equal(out_bytes[:blob_len], data), "a reconstructed blob differs from the original"
            t = d if t is None else min(t, d)
        return 1e3 * t

    r = {"log_domain": n, "blob_bytes": blob_len,
         "from_block_ms": best(lambda: lib.frieda_reconstruct_device(h, block.data_ptr(), L, n, 5, blob_len, out_bytes.data_ptr())),
         "from_points_ms": best(lambda: lib.frieda_reconstruct_points_device(h, cells.data_ptr(), idx.ctypes.data, n_pts, 0, L, n, blob_len, out_bytes.data_ptr())),
         "points": n_pts, "entry_points": "frieda_reconstruct_device (block 5 of 16), frieda_reconstruct_points_device (single points)",
         "checked": "bytes equal to the original blob; every sample equal to the re-encoded codeword (inside the call)"}
    ctx.close()
    return r


def end_to_end(frieda_amd, torch, device, n, K, cfg, expect_roots=None):
    """The reference API takes HOST bytes (`data: &[u8]`, /root/reference/src/lib.rs:31,36): the same stream of K distinct blobs
    handed over in host memory — pageable (what a Rust caller has) and page-locked — through the C ABI's throughput entry points
    (frieda_prove_many / frieda_commit_many on one device: uploads run ahead of the kernels on a copy stream), and one lone
    `frieda_commit_and_generate_proof` call.  PCIe-inclusive; never `value`."""
    blob_len = blob_len_for(n)
    seeds = [blob_len] * K
    pageable = [splitmix64_bytes(100 + i, blob_len) for i in range(K)]
    mc = frieda_amd.MultiContext([device])
    out = {"blobs": K, "blob_bytes": blob_len, "entry_points": "frieda_prove_many / frieda_commit_many (one device, host blobs), frieda_commit_and_generate_proof"}
    mc.prove_many(pageable[: min(K, 13)], seeds[: min(K, 13)], cfg)  # sizes workspaces (a full unit of four on both contexts) and the upload ring
    mc.commit_many(pageable[: min(K, 13)], 4)

    def timed(blobs):
        dtp = dtc = None
        for _ in range(2):  # the better of two passes (a first pass over freshly pinned pages has been seen to take twice as long)
            t0 = time.perf_counter()
            res = mc.prove_many(blobs, seeds, cfg)
            d1 = (time.perf_counter() - t0) / K
            roots = [r for r, _ in res]
            assert len(set(roots)) == K and all(frieda_amd.verify(p, s) for (_, p), s in zip(res, seeds)), "an end-to-end proof does not verify"
            del res
            t0 = time.perf_counter()
            croots = mc.commit_many(blobs, 4)
            d2 = (time.perf_counter() - t0) / K
            assert croots == roots
            dtp = d1 if dtp is None else min(dtp, d1)
            dtc = d2 if dtc is None else min(dtc, d2)
        if expect_roots is not None:
            assert roots[: len(expect_roots)] == expect_roots[: len(roots)], "host-blob roots differ from the device-resident run's"
        return 1e3 * dtp, 1e3 * dtc

    out["pageable_ms_per_blob"], out["pageable_commit_ms_per_blob"] = timed(pageable)
    keep = [torch.from_numpy(b).pin_memory() for b in pageable]
    out["pinned_ms_per_blob"], out["pinned_commit_ms_per_blob"] = timed([t.numpy() for t in keep])
    del keep
    mc.close()
    ctx = frieda_amd.Context(device)
    ctx.commit_and_generate_proof(pageable[0], blob_len, cfg)
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.commit_and_generate_proof(pageable[0], blob_len, cfg)
    out["lone_call_ms"] = 1e3 * (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.commit(pageable[0], 4)
    out["lone_commit_call_ms"] = 1e3 * (time.perf_counter() - t0) / reps
    ctx.close()
    elems = 4.0 * (1 << n)
    out["pageable_value"] = elems / (out["pageable_ms_per_blob"] * 1e-3)
    out["unit"] = "M31 field-elems/s"
    return out


def spm_child(args):
    """`bench.py --spm-child --gpus N`: ONE process drives all N devices through the C ABI's multi-GPU entry (frieda_multi_create,
    frieda_prove_many, frieda_commit_many: multi.cpp — what a Rust caller of /root/reference/src/lib.rs:31-38 on an 8-GPU node would
    use): host blobs, one host thread + two contexts + an upload ring per device, blob i -> device i mod N, one ncclAllGather of the
    roots per call.  Workloads: BASELINE configs[3] as written (8 blobs of the 2^22 domain, generator seeds 100 .. 107,
    /root/reference/benches/proof.rs:30-44 per blob) and N * k blobs of the headline domain (seeds 100 + j).  Prints one JSON line;
    the rank-0 process of the per-rank run merges it as `single_process_multi` and checks the roots."""
    N = args.gpus
    devices = [int(x) for x in args.spm_devices.split(",")] if args.spm_devices else list(range(N))
    out = {"devices": devices, "entry_points": "frieda_multi_create / frieda_prove_many / frieda_commit_many (one process, host blobs)"}
    if args.dry_collective:  # CPU rehearsal of the spawn / merge plumbing: no GPU, no library
        out["dry_run"] = True
        sys.stdout.write(json.dumps(out) + "\n")
        return 0
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)  # RCCL banners etc. must not land in the JSON
    out["env_defaults"] = dist_env_defaults(os.environ)  # (one process, one node: ncclCommInitAll bootstraps over the loopback)
    import frieda_amd

    cfg = frieda_amd.PcsConfig(frieda_amd.FriConfig(4, 0, 20), 20)
    t0 = time.perf_counter()
    mc = frieda_amd.MultiContext(devices)
    out["create_s"] = time.perf_counter() - t0
    out["uses_rccl"] = mc.uses_rccl
    n = args.log_domain

    def timed(blobs, seeds, passes=3):
        """best of `passes` (after one untimed call that sizes workspaces, upload rings and the communicator)"""
        mc.prove_many(blobs, seeds, cfg)
        mc.commit_many(blobs, 4)
        tp = tc = None
        for _ in range(passes):
            t0 = time.perf_counter()
            res = mc.prove_many(blobs, seeds, cfg)
            d1 = time.perf_counter() - t0
            roots = [r for r, _ in res]
            assert all(p.commitment == r and frieda_amd.verify(p, s_) for (r, p), s_ in zip(res, seeds)), "a single-process multi-GPU proof does not verify"
            del res
            t0 = time.perf_counter()
            croots = mc.commit_many(blobs, 4)
            d2 = time.perf_counter() - t0
            assert croots == roots, "frieda_commit_many and frieda_prove_many disagree on a root"
            tp = d1 if tp is None else min(tp, d1)
            tc = d2 if tc is None else min(tc, d2)
        return tp, tc, roots

    def block(log, blobs, seeds, tp, tc, roots):
        cnt, elems = len(blobs), 4.0 * (1 << log)
        return {"log_domain": log, "blobs": cnt, "blob_bytes": len(blobs[0]), "prove_ms_per_call": 1e3 * tp, "prove_ms_per_blob": 1e3 * tp / cnt,
                "value": elems * cnt / tp, "unit": "M31 field-elems/s", "frac_of_hbm_peak_wall_per_gpu": algorithmic_bytes(log, "prove") * cnt / tp / 1e9 / HBM_PEAK_GBS / len(devices),
                "commit_ms_per_call": 1e3 * tc, "commit_ms_per_blob": 1e3 * tc / cnt, "commit_value": elems * cnt / tc,
                "verified_proofs": cnt, "roots": [r.hex() for r in roots], "host_memory": "pageable", "pcie_inclusive": True}

    if n >= 22:
        # configs[3] as written: one 2^22 blob per GPU at N = 8 (two per GPU at N = 4, ...)
        blobs = [splitmix64_bytes(100 + i, blob_len_for(22)) for i in range(8)]
        seeds = [blob_len_for(22)] * 8
        c4 = block(22, blobs, seeds, *timed(blobs, seeds))
        c4["workload"] = "BASELINE configs[3]: 8 independent 2^22-domain blobs, generator seeds 100..107, blob i -> device i mod N, one root all-gather"
        gold = os.path.join(ROOT, "tests", "golden", "config4_roots.json")
        if os.path.exists(gold):
            want = json.load(open(gold))["roots"]
            assert c4["roots"] == want, "configs[3] roots differ from tests/golden/config4_roots.json (the oracle's)"
            c4["roots_equal_oracle_fixture"] = True
        out["config4"] = c4
    # the headline domain: k blobs per device
    cnt = len(devices) * max(1, args.spm_blobs_per_gpu)
    blobs = [splitmix64_bytes(100 + j, blob_len_for(n)) for j in range(cnt)]
    seeds = [blob_len_for(n)] * cnt
    hd = block(n, blobs, seeds, *timed(blobs, seeds, passes=2))
    hd["workload"] = f"{cnt} independent 2^{n}-domain blobs (generator seeds 100 + j), blob j -> device j mod N"
    out["headline"] = hd
    out["gather_count"] = mc.gather_count
    mc.close()
    os.write(real_stdout, (json.dumps(out) + "\n").encode())
    return 0


def single_process_multi(args, n_devices, expect_root_of_seed=None):
    """Spawn the --spm-child as a fresh interpreter (a child process, never an exec of this one), parse its line, compare the headline
    roots with the per-rank run's (`expect_root_of_seed(seed) -> bytes or None`).  Every failure is reported in place: this block
    must never cost the line its headline."""
    import subprocess

    argv = [sys.executable, os.path.abspath(__file__), "--spm-child", "--gpus", str(n_devices), "--log-domain", str(args.log_domain),
            "--spm-blobs-per-gpu", str(args.spm_blobs_per_gpu)]
    if args.spm_devices:
        argv += ["--spm-devices", args.spm_devices]
    if args.dry_collective:
        argv += ["--dry-collective", args.dry_collective]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_PORT", "GROUP_RANK", "ROLE_RANK",
                                                          "TORCHELASTIC_RUN_ID", "FRIEDA_BENCH_SELF_LAUNCHED", "FRIEDA_BENCH_FORCE_DIST")}
    try:
        proc = subprocess.Popen(argv, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        try:
            so, se = proc.communicate(timeout=args.spm_timeout)
        except subprocess.TimeoutExpired:
            proc.kill()  # exactly the PID started above
            proc.communicate()
            return {"error": f"--spm-child did not finish within {args.spm_timeout:.0f} s"}
        lines = [ln for ln in so.decode(errors="replace").splitlines() if ln.startswith("{")]
        if proc.returncode != 0 or len(lines) != 1:
            return {"error": f"--spm-child exited {proc.returncode}", "stderr_tail": se.decode(errors="replace")[-1500:]}
        doc = json.loads(lines[0])
        hd = doc.get("headline")
        if hd and expect_root_of_seed is not None:
            checked = 0
            for j, r in enumerate(hd["roots"]):
                want = expect_root_of_seed(100 + j)
                if want is not None:
                    if bytes.fromhex(r) != want:
                        return {"error": f"root of the seed-{100 + j} blob differs between frieda_prove_many and the per-rank run", "child": doc}
                    checked += 1
            hd["roots_equal_per_rank_run"] = checked
        for blk in ("headline", "config4"):  # the roots were compared; keep the line short
            if doc.get(blk):
                doc[blk]["roots"] = doc[blk]["roots"][:2]
        return doc
    except Exception as e:  # noqa: BLE001
        return {"error": f"{type(e).__name__}: {e}"}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--log-domain", type=int, default=24)
    ap.add_argument("--workload", choices=["prove", "commit"], default="prove")
    ap.add_argument("--cpu-sample-log", type=int, default=24, help="domain size of the CPU baseline sample (24 = the GPU workload itself, ~15 s of one core)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-twiddle-cache", action="store_true", help="regenerate twiddles every call, as the reference does")
    ap.add_argument("--batch-extra", type=int, default=4, help="blobs per call for the extra 'batched' figure (0 = skip)")
    ap.add_argument("--in-flight", type=int, default=2,
                    help="proofs (or batches) in flight in the measured loop (one context each); 1 = one at a time")
    ap.add_argument("--batch", type=int, default=0,
                    help="blobs per call in the measured loop: 0 (default) = the library's batch policy (frieda_batch_plan: workspace bytes in flight); "
                         "> 1 = about that many per call through the batched entry points; 1 = one blob per call")
    ap.add_argument("--pipeline-depth", type=int, default=None, help="deprecated alias: 0 means --in-flight 1")
    ap.add_argument("--sequential-extra", type=int, default=20, help="proofs for the extra one-at-a-time figure (0 = skip)")
    ap.add_argument("--stagger", type=int, default=0, help="experiment: 1 = the second context's first batch has half the blobs")
    ap.add_argument("--only-measured-loop", action="store_true",
                    help="profiling aid: nothing but the measured loop touches the GPU (no lone set-up steps, no instrumented replay, no extra figures), "
                         "so that every proof kernel a profiler sees is a launch of the measured loop")
    ap.add_argument("--no-by-config", action="store_true", help="skip the by_config block (the other BASELINE configurations)")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the end_to_end block (host blobs, PCIe-inclusive)")
    ap.add_argument("--no-reconstruct", action="store_true", help="skip the reconstruct block (blob back from a block / from sampled points)")
    ap.add_argument("--dry-collective", choices=["gloo"], default=None,
                    help="CPU rehearsal of the N > 1 plumbing: launcher, rendezvous, barriers, root all_gather and max-reduce over gloo; GPU work stubbed")
    ap.add_argument("--launch-timeout", type=float, default=1500.0, help="seconds the self-launcher waits for its ranks")
    ap.add_argument("--cpu-multi-child", action="store_true", help="internal: the many-core legs of the CPU baseline (forked workers, no GPU), one JSON line")
    ap.add_argument("--spm-child", action="store_true", help="internal: the single-process multi-GPU leg (frieda_prove_many over --gpus devices), one JSON line")
    ap.add_argument("--spm-devices", default=None, help="device list of the single-process leg, e.g. 0,0 (a device listed twice needs the RCCL test double)")
    ap.add_argument("--spm-blobs-per-gpu", type=int, default=8, help="headline-size blobs per device in the single-process leg (units of 1 + 4 + 3 per device, two in flight)")
    ap.add_argument("--spm-timeout", type=float, default=150.0, help="seconds the single-process leg may take (it needs ~20 s on 8 GPUs; a hung RCCL bring-up must not cost the line)")
    ap.add_argument("--no-single-process-multi", action="store_true", help="skip the single_process_multi block of an N > 1 line")
    args = ap.parse_args(argv)
    if args.only_measured_loop:
        args.no_by_config = args.no_end_to_end = args.no_cpu_baseline = args.no_reconstruct = True
        args.batch_extra = args.sequential_extra = 0
    if args.pipeline_depth is not None:
        args.in_flight = max(1, args.pipeline_depth)
    return args


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: spawn the N ranks ourselves.  This parent never imports torch.cuda or
    touches HIP (a process that has initialised the GPU must not be replaced or forked); every rank is a fresh interpreter."""
    import socket
    import subprocess

    n = args.gpus
    if not args.dry_collective:
        import torch  # device_count() does not initialise the GPU on this image: the parent stays free to spawn

        have = torch.cuda.device_count()
        if have < n:
            sys.stderr.write(f"bench.py: --gpus {n} but only {have} GPU(s) are visible (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES?): not starting any rank\n")
            return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   FRIEDA_BENCH_SELF_LAUNCHED="1")
        dist_env_defaults(env)  # (the self-launched ranks rendezvous on 127.0.0.1: one node)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=None))
    import threading

    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = time.time() + args.launch_timeout
    worst = 0
    try:
        while any(p.poll() is None for p in procs):
            if any(p.poll() not in (None, 0) for p in procs):
                break  # a rank died: the others would wait in the rendezvous or a collective for ever
            if time.time() > deadline:
                worst = 124
                break
            time.sleep(0.1)
    finally:
        for p in procs:  # exactly the PIDs started above
            if p.poll() is None:
                p.kill()
            p.wait()
            if p.returncode != 0:
                worst = max(worst, abs(p.returncode), 1)
    reader.join(timeout=10)
    out0 = b"".join(c for c in chunks if c)
    lines = [ln for ln in out0.decode(errors="replace").splitlines() if ln.startswith("{")]
    if worst == 0 and len(lines) != 1:
        worst = 1
    if lines:
        sys.stdout.write(lines[-1] + "\n")
        sys.stdout.flush()
    return worst


def dry_run(args):
    """The distributed skeleton of main() on CPU tensors over gloo: same barriers, same root all_gather, same max-reduce; the
    GPU step is a stub that fabricates a 32-byte root from (rank, step).  For tests of the N > 1 plumbing only."""
    import hashlib

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if os.environ.get("FRIEDA_BENCH_TEST_FAIL_RANK") == str(rank):  # test hook: a rank that dies before the rendezvous
        return 7
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    from frieda_amd import batch

    K = args.steps
    dist.barrier()
    t0 = time.perf_counter()
    roots = b"".join(hashlib.sha256(f"{rank}:{i}".encode()).digest() for i in range(K))
    gathered = batch.gather_rank_roots(roots, torch.device("cpu"))
    dist.barrier()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    for r in range(world):
        exp = b"".join(hashlib.sha256(f"{r}:{i}".encode()).digest() for i in range(K))
        assert bytes(gathered[r].numpy()) == exp, "root gather mismatch"
    dist.destroy_process_group()  # as in main(): nobody waits on rank 0's CPU legs
    if rank == 0:
        out = {
            "metric": "M31 field-elems/s committed (NTT+FRI+Merkle)", "value": 0.0, "unit": "M31 field-elems/s", "n_gpus": world,
            "steps": K, "warmup": args.warmup, "ms_per_step": 1e3 * float(t.item()) / max(K, 1), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "none (dry run: GPU work stubbed)", "dry_run": True,
            "config": {"workload": "dry-collective gloo: launcher + rendezvous + root all_gather only", "log_domain": args.log_domain},
            "roots_gathered": int(gathered.numel() // 32), "roofline": None, "roofline_valu": None,
        }
        if not args.no_cpu_baseline:  # the same CPU legs an N > 1 line of the real run carries (the GPU-side comparison of bench sizes needs a GPU)
            out["cpu_baseline"] = cpu_baseline(args.cpu_sample_log, args.workload, 1, all_cores_log=min(22, args.cpu_sample_log))
        else:
            out["cpu_baseline"] = None
        if not args.no_single_process_multi:
            out["single_process_multi"] = single_process_multi(args, world, None)
        sys.stdout.write(json.dumps(out) + "\n")
        sys.stdout.flush()
    return 0


def main():
    args = parse_args()
    if args.cpu_multi_child:
        sys.exit(cpu_multi_child(args))
    if args.spm_child:
        sys.exit(spm_child(args))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    if args.dry_collective:
        sys.exit(dry_run(args))
    # The contract is ONE JSON line on stdout.  RCCL (and anything else linked in) may write banners to the C stdout, which is
    # flushed at exit — after our line.  Keep the real stdout for the JSON alone and send every other fd-1 writer to stderr.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    env_defaults = dist_env_defaults(os.environ)  # reported in the line (`env_defaults`)
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start one rank per GPU (or drop WORLD_SIZE and let bench.py launch them)")
    visible = torch.cuda.device_count()
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    if visible < 1 or (visible < local_world and visible != 1):
        # (exactly one visible device per rank is the launcher-narrowed case: every rank then uses device 0)
        raise SystemExit(f"bench.py: {local_world} ranks on this node but only {visible} GPU(s) visible to rank {rank}: one GPU per rank is the contract")
    # one rank per GPU; if the launcher narrowed the visible devices per rank, index what is visible
    local_rank = local_rank % max(visible, 1)
    torch.cuda.set_device(local_rank)
    dist = None
    # FRIEDA_BENCH_FORCE_DIST=1 exercises the collective path (RCCL init, all_gather of roots, barrier, max-reduce) with a
    # single rank — the only way to validate it on a one-GPU box
    use_dist = world > 1 or os.environ.get("FRIEDA_BENCH_FORCE_DIST") == "1"
    if use_dist:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    import frieda_amd

    n = args.log_domain
    blob_len = blob_len_for(n)
    K = args.steps
    D = max(1, args.in_flight) if args.workload == "prove" else 1
    BSZ = max(0, args.batch) if args.workload == "prove" else 1  # 0: the library's batch policy decides the cut
    # K DISTINCT blobs per rank, resident in HBM before the timed region (generator: splitmix64, seeds 100 + rank * K + i; the
    # first blob of rank 0 is the seed-100 blob the CPU baseline proves, so the two roots can be compared).  One contiguous
    # [K, blob_len] array: a batch is `batch` consecutive rows.
    all_blobs = torch.empty((K, blob_len), dtype=torch.uint8, device="cuda")
    for i in range(K):
        all_blobs[i].copy_(torch.from_numpy(splitmix64_bytes(100 + rank * K + i, blob_len)))
    blobs = [all_blobs[i] for i in range(K)]
    blob = blobs[0]
    stream = torch.cuda.Stream()
    ctx = frieda_amd.Context(local_rank, stream.cuda_stream)
    if args.no_twiddle_cache:
        ctx.set_twiddle_cache(False)
    cfg = frieda_amd.PcsConfig(frieda_amd.FriConfig(4, 0, 20), 20)  # benches/proof.rs:5-12
    seed = blob_len  # benches/proof.rs:23: Some(data.len())
    roots_dev = torch.zeros(32, dtype=torch.uint8, device="cuda")
    # The blobs are independent: nothing is exchanged while they are processed.  Each rank keeps the roots of its K blobs and the
    # ranks exchange them once, after the last blob and inside the timed region: one all_gather (RCCL) of K x 32 bytes per rank.
    roots_all = torch.zeros(32 * K, dtype=torch.uint8, device="cuda")
    gathered = torch.zeros(32 * K * world, dtype=torch.uint8, device="cuda")
    # `D` proofs in flight (one context = stream + workspace each): the Fiat-Shamir latency chain of one proof runs under the
    # chip-filling kernels of the next.  D = 1 is one proof at a time.
    # `batch` > 1: the measured loop hands `batch` consecutive blobs to the batched entry points (every kernel launched once per
    # batch: the chain is paid once per batch), `D` batches in flight.
    if args.workload != "prove":
        pipe = None
    elif BSZ != 1:
        pipe = frieda_amd.BatchPipeline(local_rank, D)
    else:
        pipe = frieda_amd.ProofPipeline(local_rank, D)
    if pipe is not None and args.no_twiddle_cache:
        for c in pipe.ctxs:
            c.set_twiddle_cache(False)

    def step(i=None):  # one proof / commit at a time on `ctx` (set-up, instrumented replay, the extra figures)
        if args.workload == "prove":
            return ctx.commit_and_generate_proof_device(blob.data_ptr(), blob_len, seed, cfg)
        dst = roots_dev.data_ptr() if i is None else roots_all.data_ptr() + 32 * i
        ctx.commit_device(blobs[i if i is not None else 0].data_ptr(), blob_len, 4, dst)
        return None, None

    def batch_cut(n_blobs):
        """The blobs of a stream as calls of about BSZ blobs each, their number a multiple of the calls in flight (D) and their sizes
        equal to within one: with D = 2 an odd number of calls leaves the last one alone on the chip, its Fiat-Shamir chain and its
        launches' ramps un-overlapped (20 blobs: 4 x 5 instead of 5 x 4, ~1 % at the driver's step count; 60 blobs: 14 calls of 4 - 5)."""
        if n_blobs <= 0:
            return []
        if BSZ == 0:  # the library's policy (include/frieda_hip.h "batch policy"): the same call frieda_prove_many makes per device
            return pipe.plan(blob_len, n_blobs, cfg)
        calls = D * max(1, n_blobs // (BSZ * D))
        calls = min(calls, n_blobs)
        base, extra = divmod(n_blobs, calls)
        return [base + 1] * extra + [base] * (calls - extra)

    def run_stream(n_blobs):
        """the measured loop: blob i of this rank through the hot path, D proofs in flight; returns [(root, proof)] in order"""
        if args.workload != "prove":
            for i in range(n_blobs):
                step(i)
            return []
        out = []
        if BSZ != 1:
            i = 0
            for nb, cnt in enumerate(batch_cut(n_blobs)):
                if args.stagger and nb == 1 and D == 2 and cnt >= 2:
                    cnt = cnt // 2  # experiment: the second context starts with half a batch (the rest joins the last call)
                r = pipe.submit_device(blobs[i].data_ptr(), blob_len, blob_len, cnt, [seed] * cnt, cfg)
                if r is not None:
                    out.extend(r)
                i += cnt
            if i < n_blobs:  # (--stagger only)
                r = pipe.submit_device(blobs[i].data_ptr(), blob_len, blob_len, n_blobs - i, [seed] * (n_blobs - i), cfg)
                if r is not None:
                    out.extend(r)
        else:
            for i in range(n_blobs):
                r = pipe.submit_device(blobs[i].data_ptr(), blob_len, seed, cfg)
                if r is not None:
                    out.append(r)
        out.extend(pipe.drain())
        return out

    def gather_roots(results):
        nonlocal gathered
        if not use_dist:
            return
        from frieda_amd import batch

        if args.workload == "prove":
            roots_all.copy_(torch.frombuffer(bytearray(b"".join(r for r, _ in results)), dtype=torch.uint8))
        else:
            ctx.synchronize()  # the roots were written on the ctx stream; the collective runs on torch's
        gathered = batch.gather_rank_roots(roots_all, roots_all.device).view(-1)

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # setup, outside the contract's warm-up: the first calls size the workspace arenas, build the twiddle tables and load the
    # code objects; the chip also needs a few milliseconds of load before it settles on its clock
    if not args.only_measured_loop:
        for _ in range(2):
            step()
    run_stream(K)  # the timed run's own cut into calls: every context's workspace is sized for its largest call before the clock starts
    torch.cuda.synchronize()
    W = args.warmup
    while W > 0:  # W untimed warm-up steps through the measured loop itself
        run_stream(min(W, K))
        W -= min(W, K)
    if use_dist:
        dist.all_gather_into_tensor(gathered, roots_all)  # warm the collective (communicator set-up happens on first use)
    fence()
    if pipe is not None and hasattr(pipe, "call_latencies"):
        pipe.call_latencies.clear()
    # The interpreter's cyclic collector must not run inside the timed region: with torch imported a full collection walks ~10^6 objects
    # (40 - 65 ms), and whether one falls into the region depends on the allocation count of everything before it — a 256-blob stream of
    # the 2^20 domain read 0.27 - 0.37 ms per blob instead of 0.12 when it did (profiles/r06_gc_pause.txt).
    # The collector is switched off while the clock runs and back on after it stops; nothing the library does is affected.  (Collecting
    # right before the region instead costs ~2 ms INSIDE it: the freed pages go back to the system and the region's own host
    # allocations — proof objects, result lists — fault them in again: 1.55 vs 1.45 ms per blob at --steps 20.)
    import gc

    gc.disable()
    t0 = time.perf_counter()
    results = run_stream(K)
    gather_roots(results)
    fence()
    dt = time.perf_counter() - t0
    gc.enable()
    #  a proof's LATENCY grows with the call it rides in: _begin to the return of _finish, per call of the timed region
    call_latency = None
    pipe_phases = [c.last_prove_phases() for c in pipe.ctxs] if pipe is not None and os.environ.get("FRIEDA_BENCH_DEBUG_PHASES") else None
    if pipe is not None and getattr(pipe, "call_latencies", None):
        lat = [(c, 1e3 * t) for c, t in pipe.call_latencies]
        call_latency = {"calls": [{"blobs": c, "ms": round(t, 3)} for c, t in lat], "max_ms": max(t for _, t in lat),
                        "note": "wall time from frieda_prove_batch_begin_device to the return of frieda_prove_batch_finish of each call of the timed region "
                                "(two calls share the chip): what a blob waits for its proof in this mode; a lone call is `sequential`"}
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # every rank now holds every root: rank r's own K roots must sit in slot r
        torch.cuda.synchronize()
        gathered_host = bytes(gathered.cpu().numpy())  # rank-major: the root of generator seed 100 + r * K + i at 32 * (r * K + i)
        mine = gathered_host[32 * K * rank : 32 * K * (rank + 1)]
        assert mine == bytes(roots_all.cpu().numpy()), "root gather mismatch"
        if args.workload == "prove":
            assert mine == b"".join(r for r, _ in results)

    # correctness gate on what was just timed: EVERY one of the K proofs verifies, the K roots are distinct, and the first FRI
    # root of blob 0 equals commit()'s
    verified = None
    if args.workload == "prove":
        assert len(results) == K and len({r for r, _ in results}) == K, "the K timed blobs must give K distinct roots"
        for r, p in results:
            assert p.commitment == r and frieda_amd.verify(p, seed), "a timed proof does not verify"
        verified = K
        root, proof = results[0]
        if not args.only_measured_loop:
            ctx.commit_device(blob.data_ptr(), blob_len, 4, roots_dev.data_ptr())
            ctx.synchronize()
            assert bytes(roots_dev.cpu().numpy()) == root, "first FRI root != commit() root"
        proof0_image = proof.serialize()
        timed_roots = [r for r, _ in results]
        del results
        if pipe is not None:  # the measured loop is over: its workspaces (up to 43 GB per context) go back before the extra figures allocate theirs
            for c in pipe.ctxs:
                c.release_workspace()
    else:
        ctx.synchronize()
        root = bytes(roots_all[:32].cpu().numpy())

    host_phases = ctx.last_prove_phases() if args.workload == "prove" else None
    elems = 4.0 * (1 << n)
    value = world * elems * args.steps / dt

    # ---- instrumented replay: per-kernel HIP-event durations on the kernels' own stream ----
    kern = []
    if not args.only_measured_loop:
        ctx.set_kernel_timing(True)
        for _ in range(args.steps):
            step()
        kern = ctx.kernel_timing_report(reset=True)
        ctx.set_kernel_timing(False)
    kern.sort(key=lambda k: -k["total_ms"])
    roofline = None
    if kern:
        dom = dominant_launch(kern)
        ach = dom["alg_bytes"] / (dom["total_ms"] * 1e-3) / 1e9 if dom["total_ms"] > 0 else 0.0
        roofline = {
            "bound": "hbm",
            "kernel": dom["name"],
            "achieved": ach,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": ach / HBM_PEAK_GBS,
            "traffic": traffic_from_profiles(dom["name"], n, args.workload)[0],
            "traffic_source": traffic_from_profiles(dom["name"], n, args.workload)[1],
            "traffic_mode": "lone proof (one blob per launch), the mode of the instrumented replay behind achieved / avg_launch_us",
            "traffic_per_blob_in_measured_loop": traffic_from_profiles(dom["name"], n, args.workload)[2],
            "avg_launch_us": 1e3 * dom["total_ms"] / max(dom["launches"], 1),
            "launches_per_step": dom["launches"] / args.steps,
            "alg_bytes_per_launch": dom["alg_bytes"] / max(dom["launches"], 1),
            "measured": "HIP events on the ctx stream, instrumented replay of the timed K steps",
        }
        # what `frac` is and is not (VERDICT r05 weak #1 / task 7): a SPEED against SURVEY.md §8d's byte model, which credits every
        # logical stage's loads and stores; the kernels elide most of those stores, so the HBM system itself is far from busy
        tr = roofline["traffic"]
        roofline["frac_is"] = ("algorithmic bytes of SURVEY.md §8d (every logical stage reads its input and writes its output once, no credit for fusion) "
                               "/ measured launch time / 8 TB/s: a speed against the byte model, not HBM occupancy — the path is bound by the integer VALU work of Blake2s (roofline_valu)")
        roofline["counter_traffic_frac"] = (tr / (roofline["avg_launch_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS) if tr and roofline["avg_launch_us"] > 0 else None
        roofline["counter_traffic_over_algorithmic"] = (tr / roofline["alg_bytes_per_launch"]) if tr and roofline["alg_bytes_per_launch"] else None
        #  the family with the largest SUMMED time next to the launch the figures above are quoted on
        top = kern[0]
        roofline["top_family_by_summed_time"] = {"kernel": top["name"], "ms_per_step": top["total_ms"] / args.steps, "launches_per_step": top["launches"] / args.steps,
                                                 "frac": (top["alg_bytes"] / (top["total_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS) if top["total_ms"] > 0 else None}
    # ---- extra figure: one proof at a time (in flight 1) on blob 0 — what round 1 reported as `value` ----
    sequential = None
    if args.workload == "prove" and args.sequential_extra > 0 and world == 1:
        reps = args.sequential_extra
        step()
        torch.cuda.synchronize()
        ts0 = time.perf_counter()
        for _ in range(reps):
            r_seq, p_seq = step()
        torch.cuda.synchronize()
        dts = (time.perf_counter() - ts0) / reps
        assert r_seq == root and p_seq.serialize() == proof0_image
        sequential = {
            "in_flight": 1,
            "value": elems / dts,
            "unit": "M31 field-elems/s",
            "ms_per_proof": 1e3 * dts,
            "frac_of_hbm_peak": algorithmic_bytes(n, "prove") / dts / 1e9 / HBM_PEAK_GBS,
            "note": f"{reps} proofs of blob 0, one at a time on one context (latency of a lone proof); not the headline value",
        }
        host_phases = ctx.last_prove_phases()

    # ---- extra figure: the batched entry point (frieda_commit_and_generate_proof_batch_device): `batch` blobs of this size per
    # call, every kernel launched once for all of them, so the Fiat-Shamir / launch latency chain is paid once per batch ----
    batched = None
    if args.workload == "prove" and args.batch_extra > 1 and world == 1:
        bsz = args.batch_extra
        many = blob.repeat(bsz)
        bseeds = [seed] * bsz
        bctx = frieda_amd.Context(local_rank)
        res = bctx.commit_and_generate_proof_batch_device(many.data_ptr(), blob_len, blob_len, bsz, bseeds, cfg)  # sizes the workspace
        assert all(r == root for r, _ in res) and res[-1][1].serialize() == proof0_image
        reps = max(2, args.steps // bsz)
        torch.cuda.synchronize()
        tb0 = time.perf_counter()
        for _ in range(reps):
            res = bctx.commit_and_generate_proof_batch_device(many.data_ptr(), blob_len, blob_len, bsz, bseeds, cfg)
        torch.cuda.synchronize()
        dtb = (time.perf_counter() - tb0) / (reps * bsz)
        batched = {
            "batch": bsz,
            "value": elems / dtb,
            "unit": "M31 field-elems/s",
            "ms_per_proof": 1e3 * dtb,
            "frac_of_hbm_peak": algorithmic_bytes(n, "prove") / dtb / 1e9 / HBM_PEAK_GBS,
            "note": "`batch` blobs of the same size per call through the batched entry point; not the headline value",
        }
        del res
        # the same with two batches in flight (two contexts alternating begin / finish): chain hidden AND paid once per batch
        bctx2 = frieda_amd.Context(local_rank)
        ctxs2 = [bctx, bctx2]
        bctx2.prove_batch_begin_device(many.data_ptr(), blob_len, blob_len, bsz, bseeds, cfg)
        bctx2.prove_batch_finish(bsz)  # sizes the second workspace
        torch.cuda.synchronize()
        tb0 = time.perf_counter()
        ctxs2[0].prove_batch_begin_device(many.data_ptr(), blob_len, blob_len, bsz, bseeds, cfg)
        for i in range(1, reps):
            ctxs2[i & 1].prove_batch_begin_device(many.data_ptr(), blob_len, blob_len, bsz, bseeds, cfg)
            res = ctxs2[(i - 1) & 1].prove_batch_finish(bsz)
        res = ctxs2[(reps - 1) & 1].prove_batch_finish(bsz)
        torch.cuda.synchronize()
        dtb2 = (time.perf_counter() - tb0) / (reps * bsz)
        assert all(r == root for r, _ in res) and res[-1][1].serialize() == proof0_image
        batched["two_batches_in_flight"] = {"ms_per_proof": 1e3 * dtb2, "value": elems / dtb2,
                                            "frac_of_hbm_peak": algorithmic_bytes(n, "prove") / dtb2 / 1e9 / HBM_PEAK_GBS}
        del res
        bctx2.close()
        bctx.close()

    # ---- extra figure: twiddles regenerated on every call, as the reference does (src/commit.rs:15, src/proof.rs:47) ----
    uncached = None
    if not args.no_twiddle_cache and world == 1 and not args.only_measured_loop:
        ctx.set_twiddle_cache(False)
        reps = max(3, args.steps // 4)
        step()
        torch.cuda.synchronize()
        tu0 = time.perf_counter()
        for _ in range(reps):
            step()
        ctx.synchronize()
        dtu = (time.perf_counter() - tu0) / reps
        ctx.set_twiddle_cache(True)
        uncached = {"ms_per_step": 1e3 * dtu, "value": elems / dtu, "unit": "M31 field-elems/s",
                    "note": "same step with frieda_ctx_set_twiddle_cache(0): twiddle tables regenerated on the device every call"}

    # secondary ceiling: the path is bound by the integer VALU rate of Blake2s, not by HBM.  For the kernels that hash
    # a whole layer compare with the pure-compute compression rate of THIS device, measured here and now (frieda_ctx_blake2s_ceiling:
    # the rate depends on the clock the chip holds under the load and on the device; round 1's box gave 40.9 G leaf / 39.8 G node
    # compressions per second, profiles/r01_blake2s_rate_mi355x.txt; round 5's throughput form of the compression — runs of one rate
    # class, the wave's priority raised for its slow runs — reaches ~63 - 66 G leaf / ~60 - 62 G node, profiles/r05_blake2s_prio.txt):
    # the first tree (a proof: its leaf launch tree5_leaf; a commitment: fused with the last transform pass, whose butterflies are
    # NOT in the ideal time and which runs at 4 waves per SIMD where the ceiling kernel runs at 8, so that fraction is a lower bound
    # on the hashing efficiency) and the fused fold + tree of the first FRI layer.
    valu = None
    if args.only_measured_loop:
        ceil = {"leaf_per_s": 1.0, "node_per_s": 1.0, "node_clock_ghz": 0.0, "leaf_cycles_per_wave_compression": 0.0, "node_cycles_per_wave_compression": 0.0}
    else:
        ceil = ctx.blake2s_ceiling_ex()  # measured now, on this device, with the clock read inside the kernel (frieda_ctx_blake2s_ceiling_ex)
    leaf_rate, node_rate = ceil["leaf_per_s"], ceil["node_per_s"]

    def valu_entry(name, n_leaf, n_levels, note):
        k = next((x for x in kern if x["name"] == name), None)
        if not k or k["total_ms"] <= 0:
            return None
        n_node = sum(n_leaf / (1 << l) for l in range(1, n_levels))
        t_launch = k["total_ms"] * 1e-3 / k["launches"]
        ideal = n_leaf / leaf_rate + n_node / node_rate
        #  next to the ceiling measured with the product's own code: a pipe model that owes nothing to it — every
        # slow-class instruction (v_alignbit, v_add3) takes the SIMD 4 cycles per wave, every fast-class one is hidden behind another wave's
        # slow one; instruction counts from the generator (tools/gen_blake2s_asm.py: 350 slow per leaf, 476 per node compression; a CPU test
        # pins them), 1024 SIMDs x 64 lanes, at the nominal 2.4 GHz and at the clock the ceiling probe read in this run
        S_LEAF, S_NODE, SIMDS = 350, 476, 1024
        model_cycles = (n_leaf * S_LEAF + n_node * S_NODE) * 4.0 / 64.0 / SIMDS  # SIMD cycles the launch needs under the model
        pipe = {"model": "slow-class instruction = 4 SIMD cycles per wave, every fast-class instruction hidden behind another wave's slow one",
                "cycles_per_wave_compression": {"leaf": 4 * S_LEAF, "node": 4 * S_NODE},
                "frac_at_2.4_ghz": model_cycles / 2.4e9 / t_launch, "frac_at_probe_clock": model_cycles / (ceil["node_clock_ghz"] * 1e9) / t_launch}
        return {"kernel": name, "bound": "int32 VALU (Blake2s compression)", "achieved": (n_leaf + n_node) / t_launch / 1e9, "vs_pipe_model": pipe,
                "peak": (n_leaf + n_node) / ideal / 1e9, "unit": "G compressions/s", "frac": ideal / t_launch, "note": note,
                "clock_ghz": ceil["node_clock_ghz"], "ceiling_cycles_per_wave_compression": {"leaf": ceil["leaf_cycles_per_wave_compression"], "node": ceil["node_cycles_per_wave_compression"]},
                "peak_source": f"measured in this run on this device: {leaf_rate / 1e9:.2f} G leaf / {node_rate / 1e9:.2f} G node compressions/s "
                               f"on register-resident data at an in-kernel clock of {ceil['node_clock_ghz']:.2f} GHz (s_memtime / s_memrealtime, "
                               "frieda_ctx_blake2s_ceiling_ex), every lane chaining compressions in the product's own throughput form (runs of one "
                               "VALU rate class, the wave's priority raised for its slow runs; one generated asm block per message shape since round 6, blake2s_asm.h) "
                               "at 8 waves per SIMD: ~2130 (leaf) / ~2270 (node) SIMD cycles per wave-compression, against ~2225 / ~2330 for the pinned C++ form of "
                               "round 5, ~3950 for the scheduler's own fine interleave (rounds 1-4) and ~3150 / ~3300 with idle issue states instead of priorities"}

    if args.only_measured_loop:
        valu = {"skipped": "--only-measured-loop: neither the ceiling nor the per-kernel replay was measured"}
    elif n >= 16:
        first = valu_entry("ntt_last_tree7", float(1 << n), 7, "leaf + 6 node levels; the launch also runs 12 transform layers on 4 columns") or \
            valu_entry("tree5_leaf", float(1 << n), 5, "leaf + 4 node levels")
        fold = valu_entry("tree5_fold_circle", float(1 << (n - 1)), 5, "fold + leaf + 4 node levels of the first FRI layer") if args.workload == "prove" else None
        valu = [v for v in (first, fold) if v]

    path_bytes = algorithmic_bytes(n, args.workload)
    gpu_ms = sum(k["total_ms"] for k in kern) / args.steps

    out = {
        "metric": "M31 field-elems/s committed (NTT+FRI+Merkle)",
        "value": value,
        "unit": "M31 field-elems/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u32",
        "data": "synthetic",
        "config": {
            "workload": f"commit_and_generate_proof, 2^{n} domain" if args.workload == "prove" else f"commit, 2^{n} domain",
            "log_domain": n,
            "log_blowup_factor": 4,
            "blob_bytes": blob_len,
            "pcs_config": {"pow_bits": 20, "log_last_layer_degree_bound": 0, "n_queries": 20},
            "parallelism": f"{world} independent blobs per step, one per GPU; one all_gather of the K x 32-byte roots per rank after the last step",
            "blobs": f"{K} distinct blobs per GPU (splitmix64 seeds 100 + rank * K + i), resident in HBM; every timed proof verified after the timed region" if args.workload == "prove" else f"{K} distinct blobs per GPU",
            "in_flight": D,
            "batch": BSZ if BSZ else "library policy (frieda_batch_plan)",
            "measured_loop": ((f"the library's batch policy (frieda_batch_plan: workspace bytes in flight) — " if BSZ == 0 else f"about {BSZ} consecutive blobs per call — ") +
                              f"the {K} blobs in {len(batch_cut(K))} calls of {max(batch_cut(K))} / {min(batch_cut(K))}, a multiple of the calls in flight — (frieda_prove_batch_begin_device / _finish), {D} calls in flight" if BSZ != 1 else
                              f"one blob per call (frieda_prove_begin_device / _finish), {D} in flight") if args.workload == "prove" else "commit_device per blob, asynchronous",
            "twiddles": "regenerated per call" if args.no_twiddle_cache else "cached per context",
        },
        "roofline": roofline,
        "roofline_valu": valu,
        "seconds_per_blob": dt / args.steps,
        "input_felts_per_s": world * float(4 << (n - 4)) * args.steps / dt,
        "path": {
            "algorithmic_bytes_per_step": path_bytes,
            "achieved_GBps_wall": path_bytes / (dt / args.steps) / 1e9,
            "frac_of_hbm_peak_wall": path_bytes / (dt / args.steps) / 1e9 / HBM_PEAK_GBS,
            "gpu_kernel_ms_per_step": gpu_ms,
            "launches_per_lone_proof": sum(k["launches"] for k in kern) / args.steps,
            "launches_per_blob_in_measured_loop": sum(k["launches"] for k in kern) / args.steps / (max(batch_cut(K)) if BSZ != 1 else 1),
            "host_phase_marks_ms_last_step": host_phases,
            "frac_of_hbm_peak_kernels": (path_bytes / (gpu_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if gpu_ms > 0 else None,
            "kernels": [
                {"name": k["name"], "ms_per_step": k["total_ms"] / args.steps, "launches_per_step": k["launches"] / args.steps} for k in kern
            ],
        },
        "uncached_twiddles": uncached,
        "sequential": sequential,
        "batched": batched,
        "verified_proofs": verified,
        "root": root.hex() if root else None,
        "env_defaults": env_defaults,
        "call_latency": call_latency,
    }
    if pipe_phases:
        out["pipeline_phase_marks_ms"] = pipe_phases
    if rank == 0 and world == 1 and not args.no_end_to_end and args.workload == "prove":
        k_e2e = max(8, min(32, (K // 8) * 8))
        try:  # (an extra block must never cost the line its headline: a failure here is reported in place)
            out["end_to_end"] = end_to_end(frieda_amd, torch, local_rank, n, k_e2e, cfg, timed_roots)  # (same generator seeds: same roots)
            out["end_to_end"]["device_resident_ms_per_blob"] = 1e3 * dt / args.steps
        except AssertionError:
            raise  # wrong results are never swallowed
        except Exception as e:  # noqa: BLE001
            out["end_to_end"] = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0 and world == 1 and not args.no_reconstruct and n >= 12:
        try:
            out["reconstruct"] = reconstruct_side(frieda_amd, torch, local_rank, n)
        except AssertionError:
            raise
        except Exception as e:  # noqa: BLE001
            out["reconstruct"] = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0 and world == 1 and not args.no_by_config:
        # the other BASELINE.json configurations through the same measured loop (configs[1]-[3]: 2^20 and 2^22 domains; commit() at
        # the headline size), so that the driver-run line carries them
        rows = []
        try:
            for (cn, cw) in ((20, "prove"), (22, "prove"), (24, "prove"), (22, "commit"), (24, "commit")):
                if cn == n and cw == args.workload and BSZ == 0 and D == 2:
                    # that is `value` itself.   Like for like with BENCH_r01 .. r04, whose measured loop handed 4 blobs to every call:
                    # the same workload through that cut on this run's device (20 blobs, two calls in flight, every proof verified)
                    if cw == "prove":
                        r4 = measure_config(frieda_amd, torch, local_rank, cn, cw, 20, 4, 2, cfg)
                        out["value_fixed_batch4"] = {"measured_loop": r4["measured_loop"], "blobs": r4["blobs"], "ms_per_blob": r4["ms_per_blob"], "value": r4["value"],
                                                     "frac_of_hbm_peak_wall": r4["frac_of_hbm_peak_wall"], "verified_proofs": r4.get("verified_proofs"),
                                                     "note": "the 4-blobs-per-call cut of rounds 1 - 4: compare THIS with BENCH_r01 .. r04; `value` uses the library's batch policy"}
                    continue
                if cn > n:
                    continue  # (small test runs: nothing above the headline size)
                # blobs per stream: enough for the batch policy to reach its cut at that size (2^20: 2 calls of 256, 2^22: 2 calls of 64 —
                # a call's latency chain is ~0.5 ms whatever the size; profiles/r06_gc_pause.txt has the rate by stream length)
                kk = {(20, "prove"): 512, (22, "prove"): 128}.get((cn, cw), 64 if cn <= 22 else 20)
                rows.append(measure_config(frieda_amd, torch, local_rank, cn, cw, kk, 0 if cw == "prove" else 1, 2 if cw == "prove" else 1, cfg))
            if n >= 20:
                rows.insert(0, measure_config2(frieda_amd, torch, local_rank, 20))
        except AssertionError:
            raise
        except Exception as e:  # noqa: BLE001
            rows.append({"error": f"{type(e).__name__}: {e}"})
        out["by_config"] = rows
    # Every collective is behind us: leave the process group BEFORE rank 0's CPU legs, so that no rank waits on them (north_star wants
    # the CPU figure "in the same run" at 1, 2, 4 and 8 GPUs: an N > 1 line carries cpu_baseline, roofline and roofline_valu too).
    if use_dist:
        dist.destroy_process_group()
    if rank == 0 and not args.no_cpu_baseline:
        cb = cpu_baseline(args.cpu_sample_log, args.workload, 2 if args.cpu_sample_log <= 22 else 1, all_cores_log=min(22, args.cpu_sample_log))
        if args.cpu_sample_log == n:  # the very same blob and configuration (rank 0's first blob is the seed-100 blob): the CPU root must be the GPU root
            assert cb["root"] == out["root"], "CPU baseline root differs from the GPU root"
            cb["root_equals_gpu_root"] = True
        cb["reference_bench_sizes"] = reference_bench_sizes(ctx, frieda_amd, torch)
        out["cpu_baseline"] = cb
    elif rank == 0:
        out["cpu_baseline"] = None
    ctx.close()
    if pipe is not None:
        pipe.close()
    if rank == 0 and use_dist and args.workload == "prove" and not args.no_single_process_multi:
        # The C ABI's own multi-GPU entry (frieda_prove_many / frieda_commit_many: one process, every device, host blobs) on the same
        # node, in a fresh child process once this rank's device memory is released; the other ranks left the process group above and
        # have exited (the CPU legs take tens of seconds; the child also warms up before it times anything).
        del blobs, blob, all_blobs, roots_all, gathered
        torch.cuda.empty_cache()

        def root_of_seed(s_):
            j = s_ - 100
            return gathered_host[32 * j : 32 * j + 32] if 0 <= j < world * K else None

        out["single_process_multi"] = single_process_multi(args, world, root_of_seed)
    if rank == 0:
        os.write(real_stdout, (json.dumps(out) + "\n").encode())


if __name__ == "__main__":
    main()
