#!/bin/bash
# A/B headline only (no extra blocks): tools/ab_quick.sh <tag> <log-domain> [env assignments...] ; one line: ms per blob in the measured loop + per-kernel us of the instrumented replay
tag=$1; n=$2; shift 2
env "$@" python3 bench.py --log-domain $n --no-cpu-baseline --no-by-config --no-end-to-end --no-reconstruct --batch-extra 0 --sequential-extra 0 --steps 40 > gpurun_out/abq_$tag.json 2> gpurun_out/abq_$tag.err
python3 - <<PY
import json
try:
    d=json.load(open("gpurun_out/abq_$tag.json"))
    ks={k["name"]:round(k["ms_per_step"]*1e3,1) for k in d["path"]["kernels"]}
    v=d.get("roofline_valu") or []
    print("$tag", "ms/step", round(d["ms_per_step"],4), ks, [ (x.get("kernel"), round(x.get("frac",0),3)) for x in v] if isinstance(v,list) else "")
except Exception as e:
    print("$tag ERR", e); print(open("gpurun_out/abq_$tag.err").read()[-1500:])
PY
