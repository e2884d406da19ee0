#!/bin/bash
# lds_conflicts.sh — LDS bank-conflict counters per kernel of a lone 2^24 proof (run on the GPU box from the repo root):
# SQ_LDS_BANK_CONFLICT (extra LDS-array cycles) against SQ_LDS_IDX_ACTIVE (all LDS-array cycles), SQ_INSTS_LDS, SQ_WAIT_INST_LDS.
# usage: bash tools/lds_conflicts.sh [ENV=value ...]   e.g. FRIEDA_HIP_LIB=build_exp/oldpad/libfrieda_hip.so
export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
rm -rf gpurun_out/pmc_lds
ONE="--no-cpu-baseline --no-by-config --no-end-to-end --no-reconstruct --batch 1 --in-flight 1 --batch-extra 0 --sequential-extra 0"
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --output-format csv -d gpurun_out/pmc_lds -- python3 bench.py --steps 3 --warmup 1 $ONE > /dev/null 2> gpurun_out/pmc_lds_err.txt
python3 - <<PY
import csv, glob, collections, re
cc = glob.glob("gpurun_out/pmc_lds/**/*counter_collection.csv", recursive=True)[0]
rows = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(cc)):
    nm = re.sub(r"\(anonymous namespace\)::|frieda::k::|void ", "", r["Kernel_Name"]); nm = re.sub(r"\(.*$", "", nm)[:40]
    rows[nm][r["Counter_Name"]] += float(r["Counter_Value"])
print(f"{'kernel':42s} {'conflict/active':>15s} {'LDS insts (M)':>14s} {'wait_lds/wave_cycles':>20s}")
for nm, c in sorted(rows.items(), key=lambda x: -x[1].get("SQ_LDS_IDX_ACTIVE", 0))[:10]:
    act = c.get("SQ_LDS_IDX_ACTIVE", 0)
    if act <= 0: continue
    print(f"{nm:42s} {c.get('SQ_LDS_BANK_CONFLICT', 0) / act:15.3f} {c.get('SQ_INSTS_LDS', 0) / 1e6:14.2f} {c.get('SQ_WAIT_INST_LDS', 0) / max(c.get('SQ_WAVE_CYCLES', 1), 1):20.3f}")
PY
rm -rf gpurun_out/pmc_lds
