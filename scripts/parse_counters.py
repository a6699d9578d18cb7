"""Condense the rocprofv3 --pmc passes of scripts/profile_counters.sh into one small json: per kernel, per counter,
the median per-dispatch total (summed over the XCD/SE/instance rows of a dispatch) and the kernel's median duration."""
import collections
import csv
import glob
import json
import os
import sys


def main():
    out_dir, tag = sys.argv[1], sys.argv[2]
    want = sys.argv[3] if len(sys.argv) > 3 else ""          # substring filter on kernel names
    per = collections.defaultdict(dict)                       # kernel -> counter -> median
    dur = collections.defaultdict(list)
    for f in sorted(glob.glob(os.path.join(out_dir, "g*", "**", "*counter_collection.csv"), recursive=True)):
        acc = collections.defaultdict(float)
        names = {}
        for r in csv.DictReader(open(f)):
            acc[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
            names[r["Dispatch_Id"]] = r["Kernel_Name"]
        by = collections.defaultdict(lambda: collections.defaultdict(list))
        for (disp, c), v in acc.items():
            by[names[disp]][c].append(v)
        for k, cs in by.items():
            for c, vals in cs.items():
                vals.sort()
                per[k][c] = {"median": vals[len(vals) // 2], "dispatches": len(vals)}
    for f in sorted(glob.glob(os.path.join(out_dir, "g*", "**", "*kernel_trace.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            dur[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    summary = {"tag": tag, "kernels": {}}
    for k, cs in per.items():
        if want and want not in k:
            continue
        d = sorted(dur.get(k, [0]))
        short = k.split("(")[0][-80:]
        summary["kernels"][k] = {"median_us_under_pmc": d[len(d) // 2] / 1e3, "counters": cs}
        print("== %s  median %.1f us" % (short, d[len(d) // 2] / 1e3))
        for c in sorted(cs):
            print("   %-44s %16.0f" % (c, cs[c]["median"]))
    json.dump(summary, open(os.path.join(out_dir, "counters_%s.json" % tag), "w"), indent=1)


if __name__ == "__main__":
    main()
