#!/bin/bash
# Run ON THE GPU BOX: rocprofv3 --kernel-trace --stats of ONE bench.py run with every secondary measurement on (sorted /
# uniform / non-uniform tables, config 3, Restrict + mean over 3 x 1e6, ComputeF at 125 000 realisations in both math
# modes): average durations of every product kernel (interp1_*, interp2_kernel, restrict/mean_stage*, lift_kernel,
# evolve_kernel, ...).  Usage: scripts/profile_all_kernels.sh <tag>      output: gpurun_out/prof_<tag>/
TAG=${1:-r02_all}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
cd "$REPO"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/stats.log" 2>&1
echo "rc=$?"
ST=$(find "$OUT/stats" -name "*kernel_stats.csv" | head -1)
grep -E "anonymous namespace|Name" "$ST" | cut -c1-220
