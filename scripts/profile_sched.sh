#!/bin/bash
# Run ON THE GPU BOX: L2 hit/miss and fabric-read counters of the pipelined sweep kernel under two schedules
# (MI_SWEEP_SCHED), one rocprofv3 --pmc pass each.  Usage: scripts/profile_sched.sh <tag> <sched> [<sched> ...]
set -u
TAG=${1:-sched}; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd "$REPO"
for sc in "$@"; do
    OUT=$REPO/gpurun_out/prof_${TAG}_s$sc
    mkdir -p "$OUT"
    export MI_SWEEP_SCHED=$sc
    timeout -k 5 240 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d "$OUT/g1" -- python3 scripts/gpu_sweep_ab.py worker > "$OUT/g1.log" 2>&1
    echo "sched $sc pmc rc=$?"
    python3 scripts/parse_counters.py "$OUT" "${TAG}_s$sc" sweep_pipe | tail -25
done
