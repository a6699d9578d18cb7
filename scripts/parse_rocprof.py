"""Condense rocprofv3 CSV output (scripts/profile_bench.sh) into small summaries for profiles/."""
import collections
import csv
import glob
import json
import os
import sys


def find(d, pat):
    r = glob.glob(os.path.join(d, "**", pat), recursive=True)
    return r[0] if r else None


def kernel_durations(trace_csv):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(trace_csv)):
        d[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return d


def counters(pmc_csv):
    """kernel -> counter -> list of per-dispatch totals (summed over the per-XCD/SE rows of a dispatch)"""
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    names = {}
    for r in csv.DictReader(open(pmc_csv)):
        acc[(r["Dispatch_Id"], r["Counter_Name"])]["v"] += float(r["Counter_Value"])
        names[r["Dispatch_Id"]] = r["Kernel_Name"]
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    for (disp, cname), v in acc.items():
        out[names[disp]][cname].append(v["v"])
    return out


def main():
    out_dir, tag = sys.argv[1], sys.argv[2]
    summary = {"tag": tag, "kernels": {}}
    tr = find(os.path.join(out_dir, "stats"), "*kernel_trace.csv")
    if tr:
        for k, v in kernel_durations(tr).items():
            v.sort()
            summary["kernels"][k] = {"calls": len(v), "avg_us": sum(v) / len(v) / 1e3, "median_us": v[len(v) // 2] / 1e3,
                                     "min_us": v[0] / 1e3, "max_us": v[-1] / 1e3}
    st = find(os.path.join(out_dir, "stats"), "*kernel_stats.csv")
    if st:
        summary["kernel_stats_csv"] = open(st).read()
    pm = {}
    for sub in ("pmc_fetch", "pmc_write", "pmc_tcc"):
        f = find(os.path.join(out_dir, sub), "*counter_collection.csv")
        if not f:
            continue
        for k, cs in counters(f).items():
            for c, vals in cs.items():
                vals.sort()
                pm.setdefault(k, {})[c] = {"per_dispatch_median": vals[len(vals) // 2], "dispatches": len(vals)}
    summary["pmc"] = pm
    # HBM traffic of the interp1 kernel per launch, MI355X_MICROARCH.md section HBM: FETCH_SIZE / WRITE_SIZE are
    # in KiB; on gfx950 FETCH_SIZE reports half the bytes of wide (16 B/lane) coalesced reads -> x2.
    for k, cs in pm.items():
        if "interp1" in k and "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
            f, w = cs["FETCH_SIZE"]["per_dispatch_median"], cs["WRITE_SIZE"]["per_dispatch_median"]
            summary.setdefault("traffic", {})[k] = {
                "FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w,
                "hbm_bytes_per_launch_raw": (f + w) * 1024.0,
                "hbm_bytes_per_launch_corrected": (2.0 * f + w) * 1024.0,
                "note": "corrected = (2*FETCH_SIZE + WRITE_SIZE)*1024: gfx950 FETCH_SIZE counts 128-B read requests "
                        "as 64 B (exact for the 16 B/lane query stream; the table-gather share is uncalibrated and "
                        "may be over-corrected)"}
    # profiles/traffic_latest.json: what bench.py reports as roofline.traffic (keyed by table mode + query set)
    import re
    queries = sys.argv[3] if len(sys.argv) > 3 else "random"
    nq = int(sys.argv[4]) if len(sys.argv) > 4 else 100000000
    latest = {}
    for k, t in summary.get("traffic", {}).items():
        m = re.search(r"interp1_vec_kernel<(\d+)", k)
        if m:
            latest["interp1_mode%s_%s" % (m.group(1), queries)] = {
                "nq": nq, "hbm_bytes_per_launch": t["hbm_bytes_per_launch_corrected"],
                "hbm_bytes_per_launch_raw": t["hbm_bytes_per_launch_raw"], "kernel": k, "profile": "summary_%s.json" % tag,
                "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; (2*FETCH_SIZE + WRITE_SIZE)*1024 "
                          "(gfx950 FETCH_SIZE halves wide reads, MI355X_MICROARCH.md section HBM); Infinity-Cache hits are counted"}
    if latest:
        json.dump(latest, open(os.path.join(out_dir, "traffic_latest.json"), "w"), indent=1)
    dst = os.path.join(out_dir, "summary_%s.json" % tag)
    json.dump(summary, open(dst, "w"), indent=1)
    for k, v in summary["kernels"].items():
        print("%-90s calls=%d avg=%.1f us" % (k[:90], v["calls"], v["avg_us"]))
    print(json.dumps(summary.get("traffic", {}), indent=1))


if __name__ == "__main__":
    main()
