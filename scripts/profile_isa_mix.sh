#!/bin/bash
# Dynamic instruction mix and effective shader clock of the tile kernels (BASELINE configs[2], [3] and the doc model,
# default and profile-guided builds), on a GPU box: one kernel-trace --stats run, then separate --pmc passes (kernel trace
# only, never with other trace domains).  scripts/isa_mix_report.py merges them with the static mix of the hot loop
# (scripts/isa_mix.py) into profiles/rNN_isa_mix.json.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
RND=${1:-04}
shift
O=$R/gpurun_out/prof_isa
rm -rf $O && mkdir -p $O && cd /tmp && export TMPDIR=/tmp
export INFLX_PROBE_STAMP=$O/cases.json
P="python3 $R/scripts/isa_mix_probe.py $*"
rocprofv3 --kernel-trace --stats -d $O/stats -o t --output-format csv -- $P > $O/stats.log 2>&1 || exit 1
echo stats done
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace -d $O/pmc_a -o a -- $P > $O/pmc_a.log 2>&1 || exit 1
echo a done
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 --kernel-trace -d $O/pmc_c -o c -- $P > $O/pmc_c.log 2>&1 || exit 1
echo c done
rocprofv3 --pmc SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_SALU --kernel-trace -d $O/pmc_d -o d -- $P > $O/pmc_d.log 2>&1 || exit 1
echo d done
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_WAVES --kernel-trace -d $O/pmc_e -o e -- $P > $O/pmc_e.log 2>&1 || exit 1
echo e done
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM --kernel-trace -d $O/pmc_f -o f -- $P > $O/pmc_f.log 2>&1 || exit 1
echo f done
cd $R && python3 scripts/isa_mix_report.py profiles/r${RND}_isa_mix.json $O/cases.json $(find $O/pmc_a $O/pmc_c $O/pmc_d $O/pmc_e $O/pmc_f -name "*.db" | sort)
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $R/gpurun_out/prof_isa/kernel_stats.csv
