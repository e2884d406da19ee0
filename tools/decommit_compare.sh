#!/bin/bash
# Measurement aid: latency by size and batch throughput with the openings produced on the device (decommit.hip) and by the
# host planner + gather launch (FRIEDA_HOST_DECOMMIT=1).  usage: bash tools/decommit_compare.sh   (on the GPU box, repo root)
for mode in dev host; do
  if [ $mode = host ]; then export FRIEDA_HOST_DECOMMIT=1; else unset FRIEDA_HOST_DECOMMIT; fi
  echo "== $mode"
  SWEEP_CPU_LIMIT_S=0 python tools/size_sweep.py 2>/dev/null | cut -d'|' -f2-5
  python tools/batch_throughput.py 1024 65536 2>/dev/null | cut -d'|' -f2-5,7
done
