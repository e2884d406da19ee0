#!/bin/bash
# A/B bench lines on the GPU box: tools/ab_bench.sh <tag> <log-domain> [env assignments...] ; writes gpurun_out/ab_<tag>.json
tag=$1; n=$2; shift 2
env "$@" python3 bench.py --log-domain $n --no-cpu-baseline --batch-extra 0 --steps 40 > gpurun_out/ab_$tag.json 2> gpurun_out/ab_$tag.err
python3 - <<EOF
import json
try:
    d=json.load(open("gpurun_out/ab_$tag.json"))
    ks={k["name"]:round(k["ms_per_step"]*1e3,1) for k in d["path"]["kernels"]}
    print("$tag", "ms/step", round(d["ms_per_step"],4), "seq", round((d.get("sequential") or {}).get("ms_per_proof",0),4), "uncached", round((d.get("uncached_twiddles") or {}).get("ms_per_step",0),4), ks)
except Exception as e:
    print("$tag ERR", e); print(open("gpurun_out/ab_$tag.err").read()[-1500:])
EOF
