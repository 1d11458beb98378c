#!/bin/bash
# Wave-state counters of the IResNet-100 batch-256 forward on one lane, one pass per counter group (PMC only, no tracing).
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/waits
mkdir -p $O
W=${1:-r100}
ARGS="--workload $W --steps 2 --warmup 1 --no-cpu-baseline --no-throughput-mode"
export DIF_STREAMS=1
p() { local tag=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $O/$tag -o p -- python3 bench.py --gpus 1 $ARGS > $O/$tag.json 2> $O/$tag.err; }
p a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY &&
p b SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE &&
p c SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY &&
p d SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_VALU SQ_INSTS_SALU &&
p e SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS &&
python3 tools/pmc_table.py $O/a/p_counter_collection.csv $O/b/p_counter_collection.csv $O/c/p_counter_collection.csv $O/d/p_counter_collection.csv $O/e/p_counter_collection.csv > $O/table_$W.txt
echo "waits rc=$?"
