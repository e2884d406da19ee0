#!/bin/bash
# icache_counters.sh — instruction-cache counters per kernel of a lone 2^24 proof (run on the GPU box from the repo root): the tree kernels are
# straight-line code of ~80 KB (nine unrolled compressions), larger than the 64 KB instruction cache two CUs share.
# usage: bash tools/icache_counters.sh [ENV=value ...]
export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
rm -rf gpurun_out/pmc_ic
ONE="--no-cpu-baseline --no-by-config --no-end-to-end --no-reconstruct --batch 1 --in-flight 1 --batch-extra 0 --sequential-extra 0"
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/pmc_ic -- python3 bench.py --steps 3 --warmup 1 $ONE > /dev/null 2> gpurun_out/pmc_ic_err.txt
python3 - <<PY
import csv, glob, collections, re
cc = glob.glob("gpurun_out/pmc_ic/**/*counter_collection.csv", recursive=True)[0]
rows = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
seen = set()
for r in csv.DictReader(open(cc)):
    nm = re.sub(r"\(anonymous namespace\)::|frieda::k::|void ", "", r["Kernel_Name"]); nm = re.sub(r"\(.*$", "", nm)[:40]
    rows[nm][r["Counter_Name"]] += float(r["Counter_Value"])
    if (nm, r["Dispatch_Id"]) not in seen:
        seen.add((nm, r["Dispatch_Id"])); n[nm] += 1
print(f"{'kernel':42s} {'launches':>8s} {'icache req (M)':>14s} {'miss rate':>9s} {'dup miss':>9s} {'ifetch (M)':>10s} {'wait_inst/wave_cycles':>21s}")
for nm, c in sorted(rows.items(), key=lambda x: -x[1].get("SQ_WAVE_CYCLES", 0))[:12]:
    req = c.get("SQC_ICACHE_REQ", 0)
    if req <= 0: continue
    print(f"{nm:42s} {n[nm]:8d} {req / 1e6:14.2f} {c.get('SQC_ICACHE_MISSES', 0) / req:9.3f} {c.get('SQC_ICACHE_MISSES_DUPLICATE', 0) / req:9.3f} {c.get('SQ_IFETCH', 0) / 1e6:10.2f} {c.get('SQ_WAIT_INST_ANY', 0) / max(c.get('SQ_WAVE_CYCLES', 1), 1):21.3f}")
PY
rm -rf gpurun_out/pmc_ic
