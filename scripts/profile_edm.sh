#!/bin/bash
# Run ON THE GPU BOX (through gpurun): VALU utilisation counters of evolve_kernel (ComputeF, N = 1024, R = 32768).
# Usage: scripts/profile_edm.sh <tag> [fast]      outputs under gpurun_out/prof_edm_<tag>/
set -u
TAG=${1:-r01}
MODE=${2:-exact}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_edm_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
cd "$REPO"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 scripts/gpu_edm_profile_target.py $MODE > "$OUT/stats.log" 2>&1
echo "stats rc=$?"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --kernel-trace --output-format csv -d "$OUT/pmc_insts" -- python3 scripts/gpu_edm_profile_target.py $MODE > "$OUT/pmc_insts.log" 2>&1
echo "pmc insts rc=$?"
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/pmc_cycles" -- python3 scripts/gpu_edm_profile_target.py $MODE > "$OUT/pmc_cycles.log" 2>&1
echo "pmc cycles rc=$?"
rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU --kernel-trace --output-format csv -d "$OUT/pmc_wait" -- python3 scripts/gpu_edm_profile_target.py $MODE > "$OUT/pmc_wait.log" 2>&1
echo "pmc wait rc=$?"
python3 - "$OUT" <<'PY'
import csv, glob, sys, json, os
out = sys.argv[1]
res = {}
for d in ("pmc_insts", "pmc_cycles", "pmc_wait"):
    for f in glob.glob(os.path.join(out, d, "*", "*counter_collection.csv")):
        for row in csv.DictReader(open(f)):
            if "evolve" in row["Kernel_Name"]:
                res.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
summary = {k: sorted(v)[len(v) // 2] for k, v in res.items()}
for f in glob.glob(os.path.join(out, "stats", "*", "*kernel_stats.csv")):
    for row in csv.DictReader(open(f)):
        if "evolve" in row["Name"]:
            summary["evolve_avg_ns"] = float(row["AverageNs"]); summary["evolve_calls"] = int(row["Calls"])
json.dump(summary, open(os.path.join(out, "summary.json"), "w"), indent=1)
print(json.dumps(summary, indent=1))
PY
