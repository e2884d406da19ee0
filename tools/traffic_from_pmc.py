"""Builds profiles/<tag>_traffic.json from two rocprofv3 PMC runs of bench.py (one --pmc FETCH_SIZE, one --pmc WRITE_SIZE,
each with --kernel-trace only, as MI355X_MICROARCH.md §HBM prescribes).

Per kernel family (the names bench.py's HIP-event timer uses) and per step: FETCH_SIZE and WRITE_SIZE in KiB, and
    traffic_bytes = 1024 * (fetch_correction * FETCH_SIZE + WRITE_SIZE)
with fetch_correction = 2 where the guide's gfx950 rule applies (FETCH_SIZE reports exactly half the bytes of a wide
coalesced streaming read) — verified here on kernels whose read volume is known exactly (tree5_leaf reads 16 N bytes of
columns: FETCH_SIZE * 1024 = 8.0 N; the contiguous NTT pass reads 16 N: FETCH_SIZE * 1024 = 9.1 N incl. twiddles).
The strided NTT pass re-reads the 16.8 MB coefficient array 16 times mostly from L2/MALL, so its counter is not doubled.

usage: python tools/traffic_from_pmc.py <fetch_dir> <write_dir> <out.json>
"""
import collections
import csv
import glob
import json
import re
import sys

FAMILY = [
    (r"tree5_kernel<0, ", "tree5_leaf"),
    (r"tree5_kernel<1, ", "tree5_node"),
    (r"tree5_kernel<2, ", "tree5_fold_circle"),
    (r"tree5_kernel<3, ", "tree5_fold_line"),
    (r"tree7q_kernel", "tree7q_node"),
    (r"top_kernel", "tree_top"),
    (r"tail_kernel", "fri_tail"),
    (r"ntt_tile12_kernel<2, 4>", "ntt_pass_mid"),
    (r"ntt_tile12_kernel<3, 0>", "ntt_pass_last"),
    (r"unpack30", "unpack30"),
    (r"grind_dev_kernel", "grind"),
    (r"gather_kernel", "gather"),
]
NO_DOUBLE = {"ntt_pass_mid"}


def family(name):
    for pat, fam in FAMILY:
        if pat in name:
            return fam
    return None


def collect(d):
    f = glob.glob(d + "/*/*counter_collection.csv")[0]
    agg = collections.defaultdict(float)
    cnt = collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        fam = family(r["Kernel_Name"])
        if fam:
            agg[fam] += float(r["Counter_Value"])
            cnt[fam] += 1
    return agg, cnt


def main():
    fetch_dir, write_dir, out = sys.argv[1], sys.argv[2], sys.argv[3]
    fa, fc = collect(fetch_dir)
    wa, wc = collect(write_dir)
    res = {"_units": "averages per launch of the kernel family over the profiled run; traffic_bytes_per_launch in bytes", "kernels": {}}
    for fam in sorted(set(fa) | set(wa)):
        corr = 1.0 if fam in NO_DOUBLE else 2.0
        f, w = fa.get(fam, 0.0) / max(fc.get(fam, 0), 1), wa.get(fam, 0.0) / max(wc.get(fam, 0), 1)
        res["kernels"][fam] = {
            "launches_profiled": fc.get(fam, 0),
            "FETCH_SIZE_KiB_per_launch": f,
            "WRITE_SIZE_KiB_per_launch": w,
            "fetch_correction": corr,
            "traffic_bytes_per_launch": 1024.0 * (corr * f + w),
        }
    json.dump(res, open(out, "w"), indent=1)
    for k, v in res["kernels"].items():
        print(f"{k:20s} launches {v['launches_profiled']:4d}  traffic/launch {v['traffic_bytes_per_launch'] / 1e6:9.1f} MB")


if __name__ == "__main__":
    main()
