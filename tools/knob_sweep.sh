#!/bin/bash
# Tuning aid: bench.py (one proof at a time) under different environment knobs, three runs each.
# usage: bash tools/knob_sweep.sh "FRIEDA_T9_MAX_LOG=17 FRIEDA_TOP_MAX_LOG=9" "FRIEDA_TAIL_RUN_LOG=10" ...
for cfg in "$@"; do
  for rep in 1 2 3; do
    echo -n "$cfg: "
    env $cfg python bench.py --batch 1 --in-flight 1 --batch-extra 0 --sequential-extra 0 --no-cpu-baseline --steps 40 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4))"
  done
done
