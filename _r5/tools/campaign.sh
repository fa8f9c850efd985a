#!/bin/bash
# The GPU suite's lattice-mode / parity files under extreme settings of the scheduling knobs (tests/conftest.py: WFST_TEST_OPTIONS,
# WFST_TEST_GRAPH_OPTIONS).  Run through gpurun from the repo root:  bash tools/campaign.sh
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
FILES="tests/test_gpu_lattice.py tests/test_gpu_running_prune.py tests/test_gpu_fuzz.py tests/test_gpu_determinize.py tests/test_gpu_golden.py tests/test_gpu_parity.py tests/test_gpu_biglm.py tests/test_nbest_paths.py"
run() {
  echo "== WFST_TEST_OPTIONS=$1 WFST_TEST_GRAPH_OPTIONS=$2"
  WFST_TEST_OPTIONS="$1" WFST_TEST_GRAPH_OPTIONS="$2" timeout -s KILL 700 python -m pytest $FILES -x -q -m gpu 2>&1 | tail -2
}
run "log2_partitions=0,joint_max=6000" ""
run "log2_partitions=1,log2_lds_slots=8,joint_max=64" ""
run "channel_groups=3,use_hip_graph=0,expand_workgroups=3,insert_workgroups=5" ""
run "log2_partitions=2,expand_workgroups=7,insert_workgroups=3" "row_align_slots=1"
