#!/bin/bash
# Run ON THE GPU BOX: a few rocprofv3 --pmc passes (one small counter group each, --kernel-trace only) of
# `bench.py --config 3` -- L2 hit/miss, fabric read/write requests and bytes, wave wait cycles of the three kernels of the
# call-wide cell ordering (or of the direct kernel: pass --interp2-path direct).
# Usage: scripts/profile_config3.sh <tag> [bench.py args]      output: gpurun_out/pmc_<tag>/ + summary json
set -u
TAG=${1:-r03_c3}
shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
cd "$REPO"
ARGS="bench.py --config 3 --steps 4 --warmup 2 $*"
i=0
for g in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" \
         "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" \
         "TCC_REQ_sum TCC_BUSY_sum" "TCC_READ_sum TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
         "TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" \
         "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i + 1))
    timeout -k 5 200 rocprofv3 --pmc $g --kernel-trace --output-format csv -d "$OUT/g$i" -- python3 $ARGS > "$OUT/g$i.log" 2>&1
    echo "pass g$i ($g) rc=$?"
done
python3 scripts/parse_counters.py "$OUT" "$TAG" interp2 > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
