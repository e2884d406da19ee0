"""Latency of commit() / commit_and_generate_proof() over the reference's own bench inputs (benches/commit.rs:6-10,
benches/proof.rs:5-23: (i % 256) bytes of 1024..65536 and the 262146-byte blob) and the BASELINE.json sizes, host blob in,
proof out (PCIe included), next to the CPU oracle on this host.  Measurement aid; prints a markdown table."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import frieda_amd
from oracle import oracle as O
from conftest import pattern_bytes, splitmix64_bytes
from util import blob_len_for

ctx = frieda_amd.Context(0)
cfg = frieda_amd.PcsConfig(frieda_amd.FriConfig(4, 0, 20), 20)
ocfg = O.make_config(20, 4, 0, 20)
blob = open(os.path.join(ROOT, "tests", "golden", "blob"), "rb").read()
cases = [("pattern 1024", pattern_bytes(1024).tobytes()), ("pattern 4096", pattern_bytes(4096).tobytes()),
         ("pattern 16384", pattern_bytes(16384).tobytes()), ("pattern 65536", pattern_bytes(65536).tobytes()), ("blob 262146", blob)]
for n in (20, 22, 24):
    cases.append((f"2^{n} domain", splitmix64_bytes(1, blob_len_for(n)).tobytes()))
cpu_limit = float(os.environ.get("SWEEP_CPU_LIMIT_S", "8"))
print("| input | domain | GPU commit ms | GPU prove ms | CPU commit ms | CPU prove ms |")
print("|---|---|---|---|---|---|")
for name, data in cases:
    L = O.polynomial_from_bytes(data[:64])[1] if False else None
    seed = len(data)
    for _ in range(3):
        ctx.commit(data, 4); ctx.commit_and_generate_proof(data, seed, cfg)
    reps = 20 if len(data) < 1 << 20 else 5
    t0 = time.perf_counter()
    for _ in range(reps): root = ctx.commit(data, 4)
    tc = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps): r2, proof = ctx.commit_and_generate_proof(data, seed, cfg)
    tp = (time.perf_counter() - t0) / reps
    assert r2 == root and frieda_amd.verify(proof, seed)
    n_dom = proof.log_size_bound + 4
    est = 3e-7 * (4 << n_dom)  # rough oracle cost, to skip the very long ones
    if est < cpu_limit:
        t0 = time.perf_counter(); oroot = O.commit(data, 4); oc = time.perf_counter() - t0
        t0 = time.perf_counter(); _, op = O.commit_and_generate_proof(data, seed, ocfg); opv = time.perf_counter() - t0
        assert oroot == root and op.serialize() == proof.serialize()
        cpu = f"{1e3 * oc:.2f} | {1e3 * opv:.2f}"
    else:
        cpu = "- | -"
    print(f"| {name} | 2^{n_dom} | {1e3 * tc:.3f} | {1e3 * tp:.3f} | {cpu} |")
