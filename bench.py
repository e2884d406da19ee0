#!/usr/bin/env python3
"""bench.py — FRI-DAS commit/prove throughput on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one synthetic blob per GPU: `commit_and_generate_proof`
(/root/reference/src/proof.rs:32-77, the benches/proof.rs:30-44 workload) on a device-resident blob whose felts
exactly fill a 2^n domain at log_blowup_factor 4 (default n = 24 = BASELINE.json configs[4], the largest single-GPU
configuration): unpack -> 4 x circle NTT -> first Merkle tree -> every FRI fold + per-layer Merkle tree -> last-layer
interpolation -> grind (pow_bits 20) -> 20 query openings.  value = M31 field elements committed per second =
n_gpus * 4 * 2^n * steps / wall time, inputs already resident in HBM when the timed region starts.

Multi-GPU: one process per GPU (torch.distributed, backend nccl = RCCL), independent blobs sharded one per rank (weak
scaling); the only collective is an all_gather of the 32-byte commitment roots per step.

The JSON line also carries
  roofline     — the dominant kernel family (by summed HIP-event time on the kernels' own stream, instrumented replay of
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


def traffic_from_profiles(kernel, n, workload):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE collected
    in separate runs and corrected as MI355X_MICROARCH.md §HBM prescribes; tools/traffic_from_pmc.py).  Only valid for the
    configuration the counters were taken on (2^24 domain); None otherwise."""
    path = os.path.join(ROOT, "profiles", "r01_prove24_traffic.json")
    if n != 24 or not os.path.exists(path):
        return None
    try:
        return json.load(open(path))["kernels"][kernel]["traffic_bytes_per_launch"]
    except (KeyError, ValueError):
        return None


def cpu_baseline(sample_log, workload, calls):
    from oracle import oracle as O

    O.build()
    data = splitmix64_bytes(100, blob_len_for(sample_log))
    cfg = O.make_config(20, 4, 0, 20)
    t0 = time.perf_counter()
    root = None
    for _ in range(calls):
        if workload == "prove":
            root = O.commit_and_generate_proof(data, data.size, cfg)[0]
        else:
            root = O.commit(data, 4)
    dt = time.perf_counter() - t0
    return {
        "root": bytes(root).hex(),
        "value": 4.0 * (1 << sample_log) * calls / dt,
        "unit": "M31 field-elems/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{calls} x {'commit_and_generate_proof' if workload == 'prove' else 'commit'} on a 2^{sample_log} domain "
        f"(same generator and config as the GPU workload), {dt:.1f} s of single-thread CPU on a {os.cpu_count()}-core host; "
        "restated CPU path (oracle/), not the upstream Rust binary",
    }


def main():
    # The contract is ONE JSON line on stdout.  RCCL (and anything else linked in) may write banners to the C stdout, which is
    # flushed at exit — after our line.  Keep the real stdout for the JSON alone and send every other fd-1 writer to stderr.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
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
    ap.add_argument("--pipeline-depth", type=int, default=2, help="proofs in flight for the extra 'pipelined' figure (0 = skip)")
    args = ap.parse_args()

    # the host driver of this pool only supports dmabuf IPC: without this RCCL fails with hipIpcGetMemHandle: invalid argument
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    # one rank per GPU; if the launcher narrowed the visible devices per rank, index what is visible
    local_rank = local_rank % max(torch.cuda.device_count(), 1)
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
    blob = torch.from_numpy(splitmix64_bytes(100 + rank, blob_len)).cuda()
    stream = torch.cuda.Stream()
    ctx = frieda_amd.Context(local_rank, stream.cuda_stream)
    if args.no_twiddle_cache:
        ctx.set_twiddle_cache(False)
    cfg = frieda_amd.PcsConfig(frieda_amd.FriConfig(4, 0, 20), 20)  # benches/proof.rs:5-12
    seed = blob_len  # benches/proof.rs:23: Some(data.len())
    roots_dev = torch.zeros(32, dtype=torch.uint8, device="cuda")
    # The blobs are independent: nothing is exchanged while they are processed.  Each rank keeps the roots of its K blobs and the
    # ranks exchange them once, after the last blob and inside the timed region: one all_gather (RCCL) of K x 32 bytes per rank.
    K = args.steps
    roots_all = torch.zeros(32 * K, dtype=torch.uint8, device="cuda")
    gathered = torch.zeros(32 * K * world, dtype=torch.uint8, device="cuda")
    host_roots = []

    def step(i=None):
        if args.workload == "prove":
            root, proof = ctx.commit_and_generate_proof_device(blob.data_ptr(), blob_len, seed, cfg)
            if i is not None:
                host_roots.append(root)
            return root, proof
        dst = roots_dev.data_ptr() if i is None else roots_all.data_ptr() + 32 * i
        ctx.commit_device(blob.data_ptr(), blob_len, 4, dst)
        return None, None

    def gather_roots():
        nonlocal gathered
        if not use_dist:
            return
        from frieda_amd import batch

        if args.workload == "prove":
            roots_all.copy_(torch.frombuffer(bytearray(b"".join(host_roots)), dtype=torch.uint8))
        else:
            ctx.synchronize()  # the roots were written on the ctx stream; the collective runs on torch's
        gathered = batch.gather_rank_roots(roots_all, roots_all.device).view(-1)

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # setup, outside the contract's warm-up: the first calls size the workspace arena, build the twiddle tables and load the
    # code objects; the chip also needs a few milliseconds of load before it settles on its clock
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    last = (None, None)
    for _ in range(args.warmup):
        last = step()
    if use_dist:
        dist.all_gather_into_tensor(gathered, roots_all)  # warm the collective (communicator set-up happens on first use)
    fence()
    t0 = time.perf_counter()
    for i in range(K):
        last = step(i)
    gather_roots()
    fence()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # every rank now holds every root: rank r's own K roots must sit in slot r
        torch.cuda.synchronize()
        mine = bytes(gathered[32 * K * rank : 32 * K * (rank + 1)].cpu().numpy())
        assert mine == bytes(roots_all.cpu().numpy()), "root gather mismatch"
        if args.workload == "prove":
            assert mine == b"".join(host_roots)

    # correctness gate on what was just timed: the proof verifies and its first root equals commit()'s
    if args.workload == "prove":
        root, proof = last
        assert frieda_amd.verify(proof, seed), "timed proof does not verify"
        ctx.commit_device(blob.data_ptr(), blob_len, 4, roots_dev.data_ptr())
        ctx.synchronize()
        assert bytes(roots_dev.cpu().numpy()) == root, "first FRI root != commit() root"
    else:
        ctx.synchronize()
        root = bytes(roots_all[32 * (K - 1) :].cpu().numpy())

    host_phases = ctx.last_prove_phases() if args.workload == "prove" else None
    elems = 4.0 * (1 << n)
    value = world * elems * args.steps / dt

    # ---- instrumented replay: per-kernel HIP-event durations on the kernels' own stream ----
    ctx.set_kernel_timing(True)
    for _ in range(args.steps):
        step()
    kern = ctx.kernel_timing_report(reset=True)
    ctx.set_kernel_timing(False)
    kern.sort(key=lambda k: -k["total_ms"])
    roofline = None
    if kern:
        dom = kern[0]
        ach = dom["alg_bytes"] / (dom["total_ms"] * 1e-3) / 1e9 if dom["total_ms"] > 0 else 0.0
        roofline = {
            "bound": "hbm",
            "kernel": dom["name"],
            "achieved": ach,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": ach / HBM_PEAK_GBS,
            "traffic": traffic_from_profiles(dom["name"], n, args.workload),
            "avg_launch_us": 1e3 * dom["total_ms"] / max(dom["launches"], 1),
            "launches_per_step": dom["launches"] / args.steps,
            "alg_bytes_per_launch": dom["alg_bytes"] / max(dom["launches"], 1),
            "measured": "HIP events on the ctx stream, instrumented replay of the timed K steps",
        }
    # ---- extra figure: the same K proofs with `depth` of them in flight (one ctx = stream + workspace each) ----
    pipelined = None
    if args.workload == "prove" and args.pipeline_depth > 1 and world == 1:
        pipe = frieda_amd.ProofPipeline(local_rank, args.pipeline_depth)
        for _ in range(args.pipeline_depth + 1):
            pipe.submit_device(blob.data_ptr(), blob_len, seed, cfg)
        pipe.drain()
        torch.cuda.synchronize()
        tp0 = time.perf_counter()
        n_done = 0
        last_p = None
        for _ in range(args.steps):
            r = pipe.submit_device(blob.data_ptr(), blob_len, seed, cfg)
            if r is not None:
                n_done += 1
                last_p = r
        for r in pipe.drain():
            n_done += 1
            last_p = r
        torch.cuda.synchronize()
        dtp = time.perf_counter() - tp0
        assert n_done == args.steps and last_p[0] == root and frieda_amd.verify(last_p[1], seed)
        pipelined = {
            "depth": args.pipeline_depth,
            "value": elems * args.steps / dtp,
            "unit": "M31 field-elems/s",
            "ms_per_proof": 1e3 * dtp / args.steps,
            "frac_of_hbm_peak": algorithmic_bytes(n, "prove") / (dtp / args.steps) / 1e9 / HBM_PEAK_GBS,
            "note": "same K proofs, same blob and config, `depth` proofs in flight on separate streams; not the headline value",
        }
        pipe.close()

    # ---- extra figure: the batched entry point (frieda_commit_and_generate_proof_batch_device): `batch` blobs of this size per
    # call, every kernel launched once for all of them, so the Fiat-Shamir / launch latency chain is paid once per batch ----
    batched = None
    if args.workload == "prove" and args.batch_extra > 1 and world == 1:
        bsz = args.batch_extra
        many = blob.repeat(bsz)
        bseeds = [seed] * bsz
        bctx = frieda_amd.Context(local_rank)
        res = bctx.commit_and_generate_proof_batch_device(many.data_ptr(), blob_len, blob_len, bsz, bseeds, cfg)  # sizes the workspace
        assert all(r == root for r, _ in res) and res[-1][1].serialize() == proof.serialize()
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
        bctx.close()

    # secondary ceiling: the path is bound by the integer VALU rate of Blake2s, not by HBM.  For the first-tree
    # kernel (16.8 M leaf + 15.7 M node compressions at n = 24) compare with the chip's measured pure-compute rate
    # (profiles/r01_blake2s_rate_mi355x.txt: 40.9 G leaf / 39.8 G node compressions per second).
    valu = None
    leaf = next((k for k in kern if k["name"] == "tree5_leaf"), None)
    if leaf and leaf["total_ms"] > 0:
        n_leaf = float(1 << n)
        n_node = n_leaf * (1 / 2 + 1 / 4 + 1 / 8 + 1 / 16) if n >= 10 else 0.0
        t_launch = leaf["total_ms"] * 1e-3 / leaf["launches"]
        ideal = n_leaf / 40.9e9 + n_node / 39.8e9
        valu = {
            "kernel": "tree5_leaf",
            "bound": "int32 VALU (Blake2s compression)",
            "achieved": (n_leaf + n_node) / t_launch / 1e9,
            "peak": (n_leaf + n_node) / ideal / 1e9,
            "unit": "G compressions/s",
            "frac": ideal / t_launch,
        }

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
            "host_phase_marks_ms_last_step": host_phases,
            "frac_of_hbm_peak_kernels": (path_bytes / (gpu_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if gpu_ms > 0 else None,
            "kernels": [
                {"name": k["name"], "ms_per_step": k["total_ms"] / args.steps, "launches_per_step": k["launches"] / args.steps} for k in kern
            ],
        },
        "pipelined": pipelined,
        "batched": batched,
        "root": root.hex() if root else None,
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cb = cpu_baseline(args.cpu_sample_log, args.workload, 2 if args.cpu_sample_log <= 22 else 1)
        if args.cpu_sample_log == n:  # the very same blob and configuration: the CPU root must be the GPU root
            assert cb["root"] == out["root"], "CPU baseline root differs from the GPU root"
            cb["root_equals_gpu_root"] = True
        out["cpu_baseline"] = cb
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    ctx.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
