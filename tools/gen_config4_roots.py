"""Writes tests/golden/config4_roots.json: the commit() roots of BASELINE.json configs[3] — 8 independent 2^22-domain blobs, generator
splitmix64 seeds 100 .. 107, log_blowup_factor 4 — from the CPU oracle (oracle/, pinned to the reference's
golden root for exactly this path: codec + twiddles + circle FFT + Merkle, /root/reference/src/commit.rs:11-22,31-37).
bench.py's single-process multi-GPU leg checks frieda_prove_many / frieda_commit_many against these on the GPU box, where the
oracle's 8 x 1.5 s would otherwise sit inside the bench.     usage: python tools/gen_config4_roots.py"""
import json
import os
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import splitmix64_bytes  # noqa: E402
from oracle import oracle as O  # noqa: E402

O.build()
LOG_DOMAIN, B = 22, 4
blob_len = (4 << (LOG_DOMAIN - B)) * 30 // 8
seeds = list(range(100, 108))
with ThreadPoolExecutor(max_workers=4) as ex:
    roots = list(ex.map(lambda s: O.commit(splitmix64_bytes(s, blob_len).tobytes(), B).hex(), seeds))
doc = {"source": "oracle/frieda_oracle.c fo_commit (tools/gen_config4_roots.py); input generator splitmix64(seed), 8 LE bytes per draw",
       "log_domain": LOG_DOMAIN, "log_blowup_factor": B, "blob_bytes": blob_len, "generator_seeds": seeds, "roots": roots}
with open(os.path.join(ROOT, "tests", "golden", "config4_roots.json"), "w") as f:
    json.dump(doc, f, indent=1)
    f.write("\n")
print("\n".join(roots))
