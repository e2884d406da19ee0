import csv, sys, glob, re
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r) for r in rows)
gaps = [(ev[i+1][0] - max(e[1] for e in ev[:i+1]), i) for i in range(len(ev)-1)]
big = sorted(gaps, reverse=True)[0][1]
run = ev[big+1:]
t0 = run[0][0]; prev_end = t0
for s, e, nm, r in run:
    short = re.sub(r"\(anonymous namespace\)::|frieda::k::|void ", "", nm).split("(")[0]
    gx = r.get("Grid_Size_X") or r.get("Grid_Size"); gy = r.get("Grid_Size_Y", ""); wg = r.get("Workgroup_Size_X") or r.get("Workgroup_Size", "")
    print(f"{(s-t0)/1e3:9.1f} us  +gap {(s-prev_end)/1e3:6.1f}  dur {(e-s)/1e3:8.1f} us  {short:40s} grid {gx}x{gy} wg {wg}")
    prev_end = max(prev_end, e)
print("total", (prev_end - t0)/1e3, "us")
