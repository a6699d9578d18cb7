#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 --pmc passes (one counter group per pass, --kernel-trace only) of a
# bench.py command, for the per-CU question of VERDICT r1 item 2 (what pins the sweep kernel's gather phase).
# Usage: scripts/profile_counters.sh <tag> [bench.py args...]       outputs under gpurun_out/pmc_<tag>/
# The python program itself follows `--` (no env/bash hop: the profiler preloads into the process).
set -u
TAG=${1:-r02}
shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
cd "$REPO"
ARGS="bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extra $*"
CGROUPS=(
 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
 "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR"
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
 "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_BUSY_CU_CYCLES"
 "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum"
 "TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"
 "TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum"
 "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum"
 "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum"
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_READ_sum"
 "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum"
 "TCP_TCR_TCP_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum"
 "TCP_RFIFO_STALL_CYCLES_sum TCP_TCR_RDRET_STALL_sum"
 "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum"
 "TCP_GATE_EN2_sum TCP_TCC_WRITE_REQ_sum"
 "TCC_REQ_sum TCC_BUSY_sum"
 "TCC_READ_sum TCC_HIT_sum TCC_MISS_sum"
 "TCC_TAG_STALL_sum TCC_IB_STALL_sum"
 "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"
 "TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"
 "TD_TD_BUSY_sum TD_TC_STALL_sum"
 "TD_LOAD_WAVEFRONT_sum TD_SPI_STALL_sum"
 "GRBM_TA_BUSY GRBM_TC_BUSY"
 "FETCH_SIZE"
 "WRITE_SIZE"
)
# PMC_FROM / PMC_TO (1-based group indices) restrict the passes of one call
FROM=${PMC_FROM:-1}
TO=${PMC_TO:-${#CGROUPS[@]}}
run_pass() {   # $1 = pass name, rest = counters
    local name=$1; shift
    # a group the hardware cannot collect together makes rocprofv3 abort and then linger: bounded by timeout
    timeout -k 5 240 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/$name" -- python3 $ARGS > "$OUT/$name.log" 2>&1
    local rc=$?
    echo "pass $name ($*) rc=$rc"
    return $rc
}
i=0
for g in "${CGROUPS[@]}"; do
    i=$((i + 1))
    if [ $i -lt $FROM ] || [ $i -gt $TO ]; then continue; fi
    if ! run_pass "g$i" $g; then
        # the group did not fit the hardware counters together: one counter per pass
        j=0
        for c in $g; do j=$((j + 1)); run_pass "g${i}_$j" $c || true; done
    fi
done
python3 scripts/parse_counters.py "$OUT" "$TAG"
