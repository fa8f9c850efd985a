#!/bin/bash
# Several rocprofv3 --pmc passes (counters only, no trace domains) of one bench step, kernels enqueued one by one.
#   bash tools/pmc_passes.sh [extra bench args]    -> gpurun_out/pmc/<pass>.txt (mean per launch per kernel)
set -u
R="${GRAFT_REPO_ROOT:-$PWD}"; O="$R/gpurun_out/pmc"; rm -rf "$O"; mkdir -p "$O"; export TMPDIR=/tmp; cd "$R"
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $line --output-format csv -d "$O/p$i" -- python3 bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-service-point --no-traffic --no-hip-graph "$@" > "$O/p$i.json" 2> "$O/p$i.err"
  echo "== $line" > "$O/p$i.txt"
  python3 tools/pmc_summary.py "$O/p$i" >> "$O/p$i.txt" 2>&1
  find "$O/p$i" -name "*_counter_collection.csv" -delete
  cat "$O/p$i.txt"
done <<'LIST'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAVES
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum
TCC_ATOMIC_sum TCC_EA0_WRREQ_sum TCC_EA0_ATOMIC_sum TCC_TAG_STALL_sum
GRBM_GUI_ACTIVE GRBM_TA_BUSY
SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_INST_CYCLES_VMEM_RD SQ_LEVEL_WAVES SQ_INSTS_LDS_ATOMIC
LIST
