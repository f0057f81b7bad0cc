#!/bin/bash
# Counters of the sweep kernels, one complete_analysis sweep per example model (scripts/valu_probe.py), on a GPU box:
#   kernel-trace --stats, then separate --pmc passes (kernel trace only, never with other trace domains):
#   SQ instruction / activity counters, LDS counters, WRITE_SIZE, FETCH_SIZE.
# scripts/valu_report.py merges the databases into profiles/rNN_valu.json (stamped with the code objects' hashes).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
RND=${1:-04}
O=$R/gpurun_out/prof_tile
rm -rf $O && mkdir -p $O && cd /tmp && export TMPDIR=/tmp
export INFLX_PROBE_STAMP=$O/code_objects.json
rocprofv3 --kernel-trace --stats -d $O/stats -o t --output-format csv -- python3 $R/scripts/valu_probe.py > $O/stats.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace -d $O/pmc_a -o a -- python3 $R/scripts/valu_probe.py > $O/pmc_a.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU --kernel-trace -d $O/pmc_b -o b -- python3 $R/scripts/valu_probe.py > $O/pmc_b.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_w -o w -- python3 $R/scripts/valu_probe.py > $O/pmc_w.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_f -o f -- python3 $R/scripts/valu_probe.py > $O/pmc_f.log 2>&1 || exit 1
python3 $R/scripts/valu_report.py $R/profiles/r${RND}_valu.json $O/code_objects.json $(find $O/pmc_a $O/pmc_b $O/pmc_w $O/pmc_f -name "*.db" | sort)
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $R/profiles/r${RND}_tile_kernels_4096_kernel_stats.csv
