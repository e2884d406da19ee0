"""Builds profiles/<tag>_traffic.json from two rocprofv3 PMC runs of bench.py (one --pmc FETCH_SIZE, one --pmc WRITE_SIZE,
each with --kernel-trace only, as MI355X_MICROARCH.md §HBM prescribes).

Per kernel family (the names bench.py's HIP-event timer uses), averaged per launch: FETCH_SIZE and WRITE_SIZE in KiB, and
    traffic_bytes = 1024 * (fetch_correction * FETCH_SIZE + WRITE_SIZE)
with the correction calibrated per kernel symbol on known byte counts (RULES below), as the guide prescribes.

usage: python tools/traffic_from_pmc.py <fetch_dir | counter_collection.csv> <write_dir | counter_collection.csv> <out.json> [commit [<fetch_dir batched> <write_dir batched>]]
"""
import collections
import csv
import glob
import json
import re
import sys

# (kernel symbol substring, family, FETCH_SIZE correction).  The correction is 2 where the counter is known to report half the
# bytes (16-byte-per-lane fully coalesced streaming reads: checked on tree5_leaf, 16 N bytes of columns read, counter 8.0 N; the
# contiguous NTT pass, 16 N read, counter 9.1 N incl. twiddles; the fold kernels, 32 N' + twiddles read as two 16-byte loads per
# lane and column, counter 17.7 N') and 1 for the strided NTT pass, which re-reads the 16.8 MB coefficient array 16 times, mostly
# from L2/MALL.  (A variant of the fold kernel that read the same bytes as 16-byte loads 32 bytes apart was counted in full,
# 34 N' — the halving really is a property of the access pattern, as the guide says.)
RULES = [
    (r"tree5r_kernel<0", "tree5_leaf", 2.0),
    (r"tree5r_kernel<1", "tree5_node", 2.0),
    (r"tree5r_kernel<2", "tree5_fold_circle", 2.0),
    (r"tree5r_kernel<3", "tree5_fold_line", 2.0),
    (r"tree5_kernel<0, ", "tree5_leaf", 2.0),
    (r"tree5_kernel<1, ", "tree5_node", 2.0),
    (r"tree5_kernel<2, ", "tree5_fold_circle", 2.0),
    (r"tree5_kernel<3, ", "tree5_fold_line", 2.0),
    (r"tree9_kernel<0, ", "tree5_leaf", 2.0),
    (r"tree9_kernel<1, ", "tree5_node", 2.0),
    (r"tree9_kernel<2, ", "tree5_fold_circle", 2.0),
    (r"tree9_kernel<3, ", "tree5_fold_line", 2.0),
    (r"tree7q_kernel", "tree7q_node", 2.0),
    (r"top_kernel", "tree_top", 2.0),
    (r"tail_kernel", "fri_tail", 2.0),
    (r"ntt_tile12_kernel<2, 4>", "ntt_pass_mid", 1.0),
    # the same pass as one workgroup per SOURCE tile (default from 1024 workgroups on, round 5): the 16.8 MB of coefficients are read once,
    # 16 bytes per lane fully coalesced (the halved pattern); the pass's traffic is its 268 MB of stores either way
    (r"ntt_tile12_rep_kernel", "ntt_pass_mid", 2.0),
    (r"ntt_tile12_kernel<3, 0>", "ntt_pass_last", 2.0),
    # the fused last pass + leaf hashing reads the strided pass's output with 16-byte fully coalesced loads (the halved pattern)
    (r"ntt_last_tree_kernel", "ntt_last_tree7", 2.0),
    (r"unpack30", "unpack30", 2.0),
    (r"grind_dev_kernel", "grind", 2.0),
    (r"gather_kernel", "gather", 2.0),
    (r"decommit_kernel", "decommit", 2.0),
]


def family(name):
    for pat, fam, corr in RULES:
        if pat in name:
            return fam, corr
    return None, 1.0


def collect(d, correct):
    """per family: (sum over launches of corrected counter in KiB, launches)"""
    f = d if d.endswith(".csv") else glob.glob(d + "/*/*counter_collection.csv")[0]
    agg = collections.defaultdict(float)
    raw = collections.defaultdict(float)
    cnt = collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        fam, corr = family(r["Kernel_Name"])
        if fam:
            v = float(r["Counter_Value"])
            raw[fam] += v
            agg[fam] += v * (corr if correct else 1.0)
            cnt[fam] += 1
    return agg, raw, cnt


def csrc_sha16():
    """fingerprint of the kernel and host sources the counters were taken on (bench.py compares it with the tree it runs from)"""
    import hashlib
    import os

    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "frieda_amd", "csrc")
    h = hashlib.sha256()
    for name in sorted(os.listdir(root)):
        if name.endswith((".hip", ".h", ".cpp")):
            h.update(name.encode())
            h.update(open(os.path.join(root, name), "rb").read())
    return h.hexdigest()[:16]


def main():
    fetch_dir, write_dir, out = sys.argv[1], sys.argv[2], sys.argv[3]
    fa, fraw, fc = collect(fetch_dir, True)
    wa, _, wc = collect(write_dir, False)
    import subprocess

    try:
        commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    except OSError:
        commit = ""
    res = {"collected_at_commit": (sys.argv[4] if len(sys.argv) > 4 else commit) or None,
           "csrc_sha16": csrc_sha16(),
           "_units": "averages per launch of the kernel family over the profiled run; traffic_bytes_per_launch in bytes; "
                     "FETCH_SIZE corrected per kernel symbol (RULES in tools/traffic_from_pmc.py)",
           "mode": "lone proofs (bench.py --batch 1 --in-flight 1): one blob per launch, nothing else on the chip — the launches bench.py's "
                   "instrumented replay times", "kernels": {}}

    def table(fa, fraw, fc, wa, wc):
        t = {}
        for fam in sorted(set(fa) | set(wa)):
            nf, nw = max(fc.get(fam, 0), 1), max(wc.get(fam, 0), 1)
            f, w = fa.get(fam, 0.0) / nf, wa.get(fam, 0.0) / nw
            t[fam] = {
                "launches_profiled": fc.get(fam, 0),
                "FETCH_SIZE_KiB_per_launch": fraw.get(fam, 0.0) / nf,
                "FETCH_corrected_KiB_per_launch": f,
                "WRITE_SIZE_KiB_per_launch": w,
                "traffic_bytes_per_launch": 1024.0 * (f + w),
            }
        return t

    res["kernels"] = table(fa, fraw, fc, wa, wc)
    if len(sys.argv) > 6:  # second pair of passes: the measured loop's own mode (4 blobs per launch, 2 calls in flight)
        fb, fbraw, fbc = collect(sys.argv[5], True)
        wb, _, wbc = collect(sys.argv[6], False)
        res["batched"] = {"mode": "the measured loop with 4 blobs per call, 2 calls in flight (bench.py --only-measured-loop --batch 4): a launch covers 4 blobs",
                          "blobs_per_launch": 4, "kernels": table(fb, fbraw, fbc, wb, wbc)}
    json.dump(res, open(out, "w"), indent=1)
    for k, v in res["kernels"].items():
        b = res.get("batched", {}).get("kernels", {}).get(k)
        extra = f"   batched: {b['traffic_bytes_per_launch'] / 4e6:9.1f} MB per blob ({b['launches_profiled']} launches)" if b else ""
        print(f"{k:20s} launches {v['launches_profiled']:4d}  traffic/launch {v['traffic_bytes_per_launch'] / 1e6:9.1f} MB{extra}")


if __name__ == "__main__":
    main()
