"""Per-kernel VALU utilisation and effective clock from a rocprofv3 --pmc run of bench.py.
usage: python tools/valu_util.py <dir with *counter_collection.csv and *kernel_trace.csv>
VALUBusy% = 100 * SQ_ACTIVE_INST_VALU * 4 / 1024 SIMDs / GRBM_GUI_ACTIVE (gfx9 formula; counters summed over the 8 XCDs, GUI_ACTIVE too);
clock ~ GRBM_GUI_ACTIVE / 8 / duration: only printed for dispatches of >= 0.3 ms (the quotient reads high on shorter ones, MI355X_MICROARCH.md
"DVFS give-back"), and a profiled pass runs a few per cent below an unprofiled one; the clock the chip really holds under the hash load is read
inside a kernel by frieda_ctx_blake2s_ceiling_ex / tools/clock_probe.hip (2.39 GHz on MI355X, profiles/r03_clock_probe_mi355x.txt)."""
import collections, csv, glob, re, sys

d = sys.argv[1]
cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
dur = {}
if kt:
    for r in csv.DictReader(open(kt[0])):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
rows = collections.defaultdict(dict)
names = {}
for r in csv.DictReader(open(cc)):
    rows[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
    names[r["Dispatch_Id"]] = r["Kernel_Name"]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for did, c in rows.items():
    nm = re.sub(r"\(anonymous namespace\)::|frieda::k::|void ", "", names[did])
    nm = re.sub(r"\(.*$", "", nm)[:36]
    us = dur.get(did, 0.0)
    key = nm if us < 150 else f"{nm}@{int(round(us, -1))}us"
    a = agg[nm if us < 150 else key]
    for k, v in c.items():
        a[k] += v
    a["us"] += us
    a["n"] += 1
print(f"{'kernel':46s} {'n':>4s} {'us/launch':>10s} {'VALUbusy%':>9s} {'~clkGHz':>7s} {'wait_any%':>9s} {'wait_inst%':>10s} {'instr/wave':>10s}")
for nm, a in sorted(agg.items(), key=lambda x: -x[1]["us"]):
    if a["us"] <= 0 or a.get("GRBM_GUI_ACTIVE", 0) <= 0:
        continue
    busy = 100.0 * a.get("SQ_ACTIVE_INST_VALU", 0) * 4 / 1024 / a["GRBM_GUI_ACTIVE"]
    clk = a["GRBM_GUI_ACTIVE"] / 8 / (a["us"] * 1e-6) / 1e9
    clk_s = f"{clk:7.2f}" if a["us"] / a["n"] >= 300.0 else "      -"
    wc = max(a.get("SQ_WAVE_CYCLES", 0), 1)
    print(f"{nm:46s} {int(a['n']):4d} {a['us'] / a['n']:10.1f} {busy:9.1f} {clk_s} {100 * a.get('SQ_WAIT_ANY', 0) / wc:9.1f} {100 * a.get('SQ_WAIT_INST_ANY', 0) / wc:10.1f} {a.get('SQ_INSTS_VALU', 0) / max(a.get('SQ_WAVES', 1), 1):10.0f}")
