#!/bin/bash
# kernel traces of single proofs at the BASELINE sizes (run on the GPU box from the repo root): gpurun_out/kt<n>/
set -u
export TMPDIR=/tmp
for n in "$@"; do
  rm -rf gpurun_out/kt$n
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt$n -- python3 bench.py --log-domain $n --steps 6 --warmup 2 --no-cpu-baseline --no-by-config --no-end-to-end --no-reconstruct --batch 1 --in-flight 1 --batch-extra 0 --sequential-extra 0 > gpurun_out/kt$n.json 2> gpurun_out/kt$n.err
  python3 tools/proof_timeline.py gpurun_out/kt$n 12 > gpurun_out/timeline$n.txt 2>&1
  # keep only the per-proof text and the bench line (the raw trace is large)
  rm -rf gpurun_out/kt$n
done
