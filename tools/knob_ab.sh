#!/bin/bash
# Tuning aid: A/B of two environment settings, alternating runs (cancels box drift).  usage: bash tools/knob_ab.sh "A_VAR=.." "B_VAR=.." [rounds]
A="$1"; B="$2"; R=${3:-5}
for i in $(seq $R); do
  for cfg in "$A" "$B"; do
    echo -n "$cfg: "
    env $cfg python bench.py --batch 1 --in-flight 1 --batch-extra 0 --sequential-extra 0 --no-cpu-baseline --steps 60 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4))"
  done
done
