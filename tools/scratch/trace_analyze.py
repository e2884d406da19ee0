import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r["Grid_Size"]) if "Grid_Size" in r else int(r.get("Grid_Size_X", 0))) for r in rows]
ev.sort()
# last run = after the largest gap
gaps = [(ev[i+1][0] - max(e[1] for e in ev[:i+1]), i) for i in range(len(ev)-1)]
big = sorted(gaps, reverse=True)[:1][0][1]
run = ev[big+1:]
t0 = run[0][0]; t1 = max(e[1] for e in run)
print("span ms", (t1-t0)/1e6, "kernels", len(run))
# sweep: time with >=1 wide kernel (grid >= 2^18 threads) active
pts = []
for s, e, nm, g in run:
    w = 1 if g >= (1 << 18) else 0
    pts.append((s, 1, w)); pts.append((e, -1, -w))
pts.sort()
act = wide = 0; last = t0; t_idle = t_narrow = t_wide = 0
for t, da, dw in pts:
    dt = t - last
    if act == 0: t_idle += dt
    elif wide == 0: t_narrow += dt
    else: t_wide += dt
    act += da; wide += dw; last = t
print("idle ms", t_idle/1e6, "only-narrow ms", t_narrow/1e6, "wide ms", t_wide/1e6)
import collections
agg = collections.defaultdict(lambda: [0, 0])
for s, e, nm, g in run:
    k = nm.split("(")[0][-40:]
    agg[k][0] += 1; agg[k][1] += e - s
for k, (c, d) in sorted(agg.items(), key=lambda x: -x[1][1])[:14]:
    print(f"{k:42s} {c:4d} {d/1e6:9.3f} ms")
