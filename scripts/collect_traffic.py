"""Summarise rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE, TCC hit/miss; one pass each) for nsd_step_kernel.

usage: collect_traffic.py <fetch_dir> <write_dir> <l2_dir> <out.json> "<workload text>"
FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads,
so the corrected read traffic is 2 x FETCH_SIZE; both the raw and the corrected totals are stored.
"""
import collections, csv, glob, json, sys

def per_launch(d, kernel="nsd_step"):
    f = glob.glob(f"{d}/*/*counter_collection.csv")[0]
    agg, n = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(f)):
        if kernel in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    return {k: agg[k] / n[k] for k in agg}, max(n.values())

fetch, nf = per_launch(sys.argv[1]); write, _ = per_launch(sys.argv[2]); l2, _ = per_launch(sys.argv[3])
out = {"workload": sys.argv[5], "kernel": "nsd_step_kernel", "launches": nf,
       "FETCH_SIZE_KiB_per_launch": fetch["FETCH_SIZE"], "WRITE_SIZE_KiB_per_launch": write["WRITE_SIZE"],
       "traffic_bytes_per_launch_raw": (fetch["FETCH_SIZE"] + write["WRITE_SIZE"]) * 1024,
       "traffic_bytes_per_launch": (2 * fetch["FETCH_SIZE"] + write["WRITE_SIZE"]) * 1024,
       "l2_hit_rate": l2["TCC_HIT_sum"] / (l2["TCC_HIT_sum"] + l2["TCC_MISS_sum"])}
json.dump(out, open(sys.argv[4], "w"), indent=1)
print(out)
