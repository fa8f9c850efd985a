#!/bin/bash
# The two SQ counter passes of tools/pmc_passes.sh on one configuration (one channel group, one step, kernels enqueued one by one):
# waves, busy and wait cycles, instruction mix per kernel -> gpurun_out/sq/<name>_p{1,2}.txt.   bash tools/sq_passes.sh NAME [bench args]
set -u
R="${GRAFT_REPO_ROOT:-$PWD}"; name="$1"; shift; O="$R/gpurun_out/sq"; mkdir -p "$O"; export TMPDIR=/tmp; cd "$R"
i=0
for line in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAVES" \
            "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
            "SQ_LEVEL_WAVES SQ_INST_LEVEL_VMEM SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $line --output-format csv -d "$O/${name}_p$i" -- python3 bench.py --groups 1 --steps 1 --warmup 0 --cpu-sample 0 --no-service-point --no-traffic --no-legs --no-hip-graph --no-profile-step "$@" > "$O/${name}_p$i.json" 2> "$O/${name}_p$i.err"
  echo "== $line" > "$O/${name}_p$i.txt"
  python3 tools/pmc_summary.py "$O/${name}_p$i" >> "$O/${name}_p$i.txt" 2>&1
  rm -rf "$O/${name}_p$i"
  cat "$O/${name}_p$i.txt"
done
