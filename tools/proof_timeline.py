"""Prints the launch timeline of ONE proof from a `rocprofv3 --kernel-trace` run of bench.py: per kernel its start offset,
duration and the idle gap before it.  usage: python tools/proof_timeline.py <trace-dir> [which-proof-from-the-end]"""
import csv, glob, sys

d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/kt"
which = int(sys.argv[2]) if len(sys.argv) > 2 else 2
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))), key=lambda x: x[0])
# a proof starts at each unpack30 launch
starts = [i for i, r in enumerate(rows) if "unpack30" in r[2]]
i0 = starts[-which]
i1 = starts[-which + 1] if which > 1 else len(rows)
t0 = rows[i0][0]
prev_end = t0
tot_gap = tot_busy = 0.0
for s, e, name in rows[i0:i1]:
    import re

    short = re.sub(r"\(anonymous namespace\)::|frieda::k::|void ", "", name)
    short = re.sub(r"\(.*$", "", short)[:40]
    gap = (s - prev_end) / 1e3
    tot_gap += max(gap, 0)
    tot_busy += (e - s) / 1e3
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  gap {gap:6.1f}  {short}")
    prev_end = max(prev_end, e)
print(f"span {(prev_end - t0) / 1e3:.1f} us, busy {tot_busy:.1f}, gaps {tot_gap:.1f}, launches {i1 - i0}")
# idle time between consecutive proofs: end of a proof's last kernel -> start of the next proof's first kernel
gaps = []
for a, b in zip(starts[:-1], starts[1:]):
    last_end = max(r[1] for r in rows[a:b])
    gaps.append((rows[b][0] - last_end) / 1e3)
if gaps:
    print("between proofs (us):", " ".join(f"{g:.0f}" for g in gaps[-8:]))
