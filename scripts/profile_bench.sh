#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 kernel-trace stats + separate PMC passes for bench.py.
# Usage: scripts/profile_bench.sh <round-tag> [random|sorted|uniform] [extra bench.py args, e.g. --config 3 | --table nonuniform]
# outputs under gpurun_out/prof_<tag>/
# The python program itself follows `--` (no env/bash hop: the profiler preloads into the process).
set -u
TAG=${1:-r01}
QUERIES=${2:-random}        # random | sorted | uniform (bench.py --queries)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
# a previous run's CSVs under the same tag must not be picked up by the parser (round 2 committed a stale kernel_stats.csv
# that way): every pass starts from an empty directory
rm -rf "$OUT/stats" "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/pmc_tcc" "$OUT/pmc_tcc2" "$OUT/pmc_tcc3" "$OUT/pmc_clk"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
cd "$REPO"
shift; shift
ARGS="bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --queries $QUERIES $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 $ARGS > "$OUT/stats.log" 2>&1
echo "stats rc=$?"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -- python3 $ARGS > "$OUT/pmc_fetch.log" 2>&1
echo "pmc fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -- python3 $ARGS > "$OUT/pmc_write.log" 2>&1
echo "pmc write rc=$?"
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-trace --output-format csv -d "$OUT/pmc_tcc" -- python3 $ARGS > "$OUT/pmc_tcc.log" 2>&1
echo "pmc tcc rc=$?"
# size classes of the L2's fabric reads: the bytes that really left L2 (FETCH_SIZE tallies every request at 64 B)
rocprofv3 --pmc TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --kernel-trace --output-format csv -d "$OUT/pmc_tcc2" -- python3 $ARGS > "$OUT/pmc_tcc2.log" 2>&1
echo "pmc tcc2 rc=$?"
# the L2's own load: requests it served and the cycles its channels were busy; GPU clock cycles of the launch (-> clock rate)
rocprofv3 --pmc TCC_REQ_sum TCC_BUSY_sum --kernel-trace --output-format csv -d "$OUT/pmc_tcc3" -- python3 $ARGS > "$OUT/pmc_tcc3.log" 2>&1
echo "pmc tcc3 rc=$?"
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/pmc_clk" -- python3 $ARGS > "$OUT/pmc_clk.log" 2>&1
echo "pmc clk rc=$?"
python3 scripts/parse_rocprof.py "$OUT" "$TAG" "$QUERIES"
