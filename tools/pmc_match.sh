#!/bin/bash
# Wave-state counters of dif_match (tools/match_prof.py), one pass per counter group (PMC only, no tracing).
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/match_pmc
mkdir -p $O
ARGS="${1:-1000000} ${2:-512} 6"
p() { local tag=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $O/$tag -o p -- python3 tools/match_prof.py $ARGS > $O/$tag.log 2>&1; }
p a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY &&
p b SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE &&
p c SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY &&
p e SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS &&
python3 tools/pmc_table.py $O/a/p_counter_collection.csv $O/b/p_counter_collection.csv $O/c/p_counter_collection.csv $O/e/p_counter_collection.csv > $O/table.txt
echo "pmc rc=$?"
