#!/bin/bash
# first_multi_gpu_run.sh — the first run on a node with more than one MI355X, as ONE command (VERDICT r05 task 4a).
# No N > 1 hardware run has happened in six rounds (the pool hands out one-GPU boxes): everything multi-GPU is rehearsed against an RCCL
# test double and gloo.  Whoever gets a node runs this from the repository root; it needs no network and writes under gpurun_out/multi/.
#   1. the GPU tests that skip below two visible GPUs (real RCCL root gather from C++ and Python, BASELINE configs[3] as written)
#   2. bench.py --gpus 1 / 2 / 4 / 8 --steps 20, each in a fresh process (one rank per GPU over RCCL; bench.py starts its own ranks)
#   3. the single-process path over all GPUs (frieda_prove_many / frieda_commit_many: one process, ncclCommInitAll)
# and prints, per N: ranks RCCL saw, roots equal on every rank, aggregate M31 elems/s, efficiency against N = 1.
# Every step runs under its own timeout and the script stops at the first GPU step that fails or times out (no retries).
# usage: bash tools/first_multi_gpu_run.sh [max_gpus]     (default: every visible GPU, at most 8)
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/multi
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=${HSA_ENABLE_IPC_MODE_LEGACY:-0}   # this pool's host driver only supports dmabuf IPC
HAVE=$(python3 -c 'import torch; print(torch.cuda.device_count())')
MAX=${1:-$HAVE}; [ "$MAX" -gt 8 ] && MAX=8; [ "$MAX" -gt "$HAVE" ] && MAX=$HAVE
echo "visible GPUs: $HAVE, using up to $MAX"
if [ "$HAVE" -lt 2 ]; then echo "this box has fewer than two GPUs: nothing here that the one-GPU runs do not already cover"; exit 3; fi
python3 -c 'import __graft_entry__ as g; g.build()' > $OUT/build.log 2>&1 || { echo "build failed: $OUT/build.log"; exit 1; }

echo "== 1. GPU tests that need >= 2 GPUs (real RCCL)"
timeout -k 10 1500 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -k "real_rccl or config4 or multi" > $OUT/tests.log 2>&1
rc=$?; tail -3 $OUT/tests.log
[ $rc -ne 0 ] && { echo "multi-GPU tests failed (rc $rc): $OUT/tests.log"; exit 1; }

echo "== 2. bench.py, one rank per GPU"
for N in 1 2 4 8; do
  [ "$N" -gt "$MAX" ] && continue
  timeout -k 10 1500 python3 bench.py --gpus $N --steps 20 --warmup 3 > $OUT/bench_n$N.json 2> $OUT/bench_n$N.err
  rc=$?
  if [ $rc -ne 0 ]; then echo "bench.py --gpus $N failed (rc $rc): $OUT/bench_n$N.err"; tail -5 $OUT/bench_n$N.err; exit 1; fi
done

echo "== 3. one process over all $MAX GPUs (frieda_prove_many / frieda_commit_many)"
timeout -k 10 600 python3 bench.py --spm-child --gpus $MAX --log-domain 24 > $OUT/spm.json 2> $OUT/spm.err
rc=$?; [ $rc -ne 0 ] && { echo "single-process leg failed (rc $rc): $OUT/spm.err"; tail -5 $OUT/spm.err; exit 1; }

python3 - "$OUT" "$MAX" <<'PY'
import json, os, sys
out, mx = sys.argv[1], int(sys.argv[2])
base = None
print(f"{'N':>2} {'elems/s':>12} {'ms/step':>9} {'eff vs N=1':>10}  checks")
for n in (1, 2, 4, 8):
    p = os.path.join(out, f"bench_n{n}.json")
    if n > mx or not os.path.exists(p):
        continue
    d = json.loads(open(p).read().strip().splitlines()[-1])
    if n == 1:
        base = d["value"]
    eff = d["value"] / (n * base) if base else float("nan")
    checks = [f"n_gpus={d['n_gpus']}", f"verified_proofs={d.get('verified_proofs')}", f"env={d.get('env_defaults')}"]
    spm = d.get("single_process_multi")
    if isinstance(spm, dict):
        hd = spm.get("headline") or {}
        checks.append(f"single_process_multi: rccl={spm.get('uses_rccl')} roots_equal_per_rank_run={hd.get('roots_equal_per_rank_run')} err={spm.get('error')}")
    print(f"{n:>2} {d['value']:>12.4g} {d['ms_per_step']:>9.4f} {eff:>10.3f}  " + "; ".join(checks))
s = json.loads(open(os.path.join(out, "spm.json")).read().strip().splitlines()[-1])
print("single process:", {k: (v if not isinstance(v, dict) else {a: b for a, b in v.items() if a != "roots"}) for k, v in s.items()})
print("(bench.py asserts inside every run: each rank's K roots sit in its slot of the gathered buffer, every timed proof verifies, rank 0's first root equals the CPU oracle's)")
PY
