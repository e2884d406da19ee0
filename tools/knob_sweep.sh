#!/bin/bash
# Tuning aid: bench.py under different kernel-choice knobs (FRIEDA_T9_MAX_LOG, FRIEDA_TOP_MAX_LOG).  usage: bash tools/knob_sweep.sh "17 9" "19 10" ...
for cfg in "$@"; do
  set -- $cfg
  for rep in 1 2 3; do
    echo -n "T9=$1 TOP=$2: "
    FRIEDA_T9_MAX_LOG=$1 FRIEDA_TOP_MAX_LOG=$2 python bench.py --pipeline-depth 0 --batch-extra 0 --no-cpu-baseline --steps 40 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4))"
  done
done
