"""Condense rocprofv3 CSV output (scripts/profile_bench.sh) into small summaries for profiles/."""
import collections
import csv
import glob
import json
import os
import sys


def find(d, pat):
    """the NEWEST file under d that matches (a directory that was not cleared between runs may hold several)"""
    r = glob.glob(os.path.join(d, "**", pat), recursive=True)
    return max(r, key=os.path.getmtime) if r else None


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
        # the stand-alone copy for profiles/ comes from the SAME file the summary quotes
        open(os.path.join(out_dir, "kernel_stats_%s.csv" % tag), "w").write(summary["kernel_stats_csv"])
    pm = {}
    for sub in ("pmc_fetch", "pmc_write", "pmc_tcc", "pmc_tcc2", "pmc_tcc3", "pmc_clk"):
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
    # the dominant interp1 kernel of the run (region-sweep or streaming), by average duration
    dom = [k for k in summary["kernels"] if "interp1_sweep_kernel" in k or "interp1_vec_kernel" in k or "interp2_kernel" in k or "interp1_sweep_pipe_kernel" in k]
    dom = max(dom, key=lambda k: summary["kernels"][k]["avg_us"] * summary["kernels"][k]["calls"]) if dom else None   # by total time
    queries = sys.argv[3] if len(sys.argv) > 3 else "random"
    nq = int(sys.argv[4]) if len(sys.argv) > 4 else 100000000
    for k, cs in pm.items():
        if k == dom and "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
            f, w = cs["FETCH_SIZE"]["per_dispatch_median"], cs["WRITE_SIZE"]["per_dispatch_median"]
            # calibration on the known byte count of this access pattern: the query stream is nq*8 bytes read as
            # 128-B requests (tallied at 64 B: half of it is missing from FETCH_SIZE); what is left of FETCH_SIZE
            # is table-gather misses, 64-B requests tallied in full (TCC_EA0_RDREQ and TCC_MISS agree, see README)
            stream = nq * 8.0
            # size classes of the fabric reads, when collected: every request at its real size (round 2: all 14.1 M
            # requests of the sweep kernel are 128-B requests, also the table-gather misses -- the "calibrated" figure
            # of round 1 took those for 64 B and was 0.5 GB short)
            sized = None
            if "TCC_EA0_RDREQ_128B_sum" in cs and "TCC_EA0_RDREQ_sum" in cs:
                n128 = cs["TCC_EA0_RDREQ_128B_sum"]["per_dispatch_median"]
                n64 = cs.get("TCC_EA0_RDREQ_64B_sum", {"per_dispatch_median": 0.0})["per_dispatch_median"]
                n32 = cs.get("TCC_EA0_RDREQ_32B_sum", {"per_dispatch_median": 0.0})["per_dispatch_median"]
                sized = 128.0 * n128 + 64.0 * n64 + 32.0 * n32 + w * 1024.0
            summary.setdefault("traffic", {})[k] = {
                "hbm_bytes_per_launch_by_request_size": sized,
                "FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w,
                "hbm_bytes_per_launch_raw": (f + w) * 1024.0,
                "hbm_bytes_per_launch_doubled": (2.0 * f + w) * 1024.0,
                "hbm_bytes_per_launch_calibrated": f * 1024.0 + 0.5 * stream + w * 1024.0,
                "table_gather_bytes_beyond_l2": (sized - w * 1024.0 - stream) if sized else max(0.0, f * 1024.0 - 0.5 * stream),
                "note": "doubled = (2*FETCH_SIZE + WRITE_SIZE)*1024 (every read taken as a 128-B request tallied at "
                        "64 B); calibrated = FETCH_SIZE*1024 + nq*4 + WRITE_SIZE*1024 (only the 16 B/lane query "
                        "stream is wide; table-gather misses are 64-B requests counted in full)"}
    # profiles/traffic_latest.json: what bench.py reports as roofline.traffic (keyed by table mode + query set)
    import re
    latest = {}
    for k, t in summary.get("traffic", {}).items():
        m = re.search(r"interp1_(?:vec|sweep_pipe|sweep)_kernel<(\d+)", k)
        if m:
            latest["interp1_mode%s_%s" % (m.group(1), queries)] = {
                "nq": nq,
                "hbm_bytes_per_launch": t["hbm_bytes_per_launch_by_request_size"] or t["hbm_bytes_per_launch_doubled"],
                "hbm_bytes_per_launch_calibrated_r01": t["hbm_bytes_per_launch_calibrated"],
                "hbm_bytes_per_launch_doubled": t["hbm_bytes_per_launch_doubled"],
                "hbm_bytes_per_launch_raw": t["hbm_bytes_per_launch_raw"],
                "table_gather_bytes_beyond_l2": t["table_gather_bytes_beyond_l2"], "kernel": k,
                "profile": "summary_%s.json" % tag,
                "method": "rocprofv3 --pmc passes, one counter group each (MI355X_MICROARCH.md section HBM: gfx950 FETCH_SIZE "
                          "tallies every fabric read at 64 B): reads = 128*TCC_EA0_RDREQ_128B + 64*_64B + 32*_32B, writes = "
                          "WRITE_SIZE*1024; without the size classes the 'doubled' figure (2*FETCH_SIZE + WRITE_SIZE)*1024; "
                          "Infinity-Cache hits are counted"}
        m2 = re.search(r"interp2_kernel<", k)
        if m2:
            latest["interp2_%s" % queries] = {
                "nq": nq, "hbm_bytes_per_launch": t["hbm_bytes_per_launch_by_request_size"] or t["hbm_bytes_per_launch_doubled"],
                "kernel": k, "profile": "summary_%s.json" % tag, "family": "interp2",
                "method": "reads = 128*TCC_EA0_RDREQ_128B + 64*_64B + 32*_32B, writes = WRITE_SIZE*1024 (separate rocprofv3 --pmc passes)"}
    # the L2 request roofline of the dominant kernel: requests served per launch / (128 channels x one request per clock)
    if dom and dom in pm and "TCC_REQ_sum" in pm[dom]:
        req = pm[dom]["TCC_REQ_sum"]["per_dispatch_median"]
        dur_us = summary["kernels"][dom]["median_us"]
        clk = None
        if "GRBM_GUI_ACTIVE" in pm[dom]:
            # summed over the eight XCD instances by counters(); a launch keeps every XCD busy for its whole duration
            c = pm[dom]["GRBM_GUI_ACTIVE"]["per_dispatch_median"] / 8.0 / (dur_us * 1e-6)
            clk = c if 1.5e9 < c < 2.7e9 else None
        clk = clk or 2.4e9
        l2 = {"tcc_req_per_launch": req, "gpu_clock_hz": clk, "l2_channels": 128,
              "l2_request_bound_ms": req / (128.0 * clk) * 1e3,
              "tcc_busy_frac": (pm[dom]["TCC_BUSY_sum"]["per_dispatch_median"] / (128.0 * clk * dur_us * 1e-6)) if "TCC_BUSY_sum" in pm[dom] else None}
        # the rate an XCD's vector request path was measured to sustain (not re-measured here: the experiment's own log)
        l2["request_cap_per_xcd_per_ns"] = 33.0
        l2["request_cap_source"] = ("profiles/r03_exp_gather_rate_vs_active_cus.log: all 32 CUs of an XCD gathering from an "
                                    "L2-resident 2 MiB table, 1.03 distinct lines per ns per CU")
        summary["l2_request_roofline"] = {dom: l2}
        for e in latest.values():
            if e["kernel"] == dom:
                e.update(l2)
    if latest and st:
        # the kernel the traffic figure belongs to must be the dominant kernel of the kernel-stats CSV committed with it
        rows = list(csv.DictReader(open(st)))
        top = max(rows, key=lambda r: float(r["TotalDurationNs"]))["Name"]
        for e in latest.values():
            assert e["kernel"] == top, "traffic_latest names %r but the dominant kernel of %s is %r" % (e["kernel"], st, top)
    if latest:
        # stamp: the sources the profiled library was built from (bench.py quotes the figure only for the same build)
        try:
            sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
            from armadillocudalinearinterpolation_amd import _build
            stamp = _build.source_hash("interp1")
        except Exception:
            stamp = None
        for e in latest.values():
            e["source_sha256"] = _build.source_hash(e.get("family", "interp1")) if stamp else None
        json.dump(latest, open(os.path.join(out_dir, "traffic_latest.json"), "w"), indent=1)
    dst = os.path.join(out_dir, "summary_%s.json" % tag)
    json.dump(summary, open(dst, "w"), indent=1)
    for k, v in summary["kernels"].items():
        print("%-90s calls=%d avg=%.1f us" % (k[:90], v["calls"], v["avg_us"]))
    print(json.dumps(summary.get("traffic", {}), indent=1))


if __name__ == "__main__":
    main()
