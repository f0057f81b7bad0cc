#!/bin/bash
# (GPU box) Dynamic VALU counts and cycles of the tile kernel for the IEEE-division build (hoist_reciprocals=False) against Compiler(hoist_reciprocals="inline"):
# three separate --pmc passes (kernel trace only) over `scripts/hoist_experiment.py MODEL:nohoist MODEL:inline`, 8 parameter rows per dispatch.
# usage: inline_counters.sh MODEL   -> gpurun_out/inline_counters_MODEL/{a,b,c}/*counter_collection.csv (dispatches in launch order:
# first the IEEE build's, then the inline build's)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
M=${1:-egno}
O=$R/gpurun_out/inline_counters_$M
rm -rf $O && mkdir -p $O && cd /tmp && export TMPDIR=/tmp
export INFLX_EXPERIMENT_P=8
P="python3 $R/scripts/hoist_experiment.py $M:nohoist $M:inline"
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace -d $O/a -o a --output-format csv -- $P > $O/a.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 --kernel-trace -d $O/b -o b --output-format csv -- $P > $O/b.log 2>&1 || exit 1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace -d $O/c -o c --output-format csv -- $P > $O/c.log 2>&1 || exit 1
$P > $O/plain.log 2>&1
tail -4 $O/plain.log
