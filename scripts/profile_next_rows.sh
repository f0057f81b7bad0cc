#!/bin/bash
# rocprofv3 kernel statistics of the SURVEY section 8(f) kernels (single-quantity sweeps, raw / Hesse planes, trajectory) -> profiles/rNN_next_rows_*
set -o pipefail
RND=${1:-05}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_next_rows
rm -rf $O && mkdir -p $O && cd /tmp && export TMPDIR=/tmp
export INFLX_EXPERIMENT_ROUNDS=1
rocprofv3 --kernel-trace --stats -d $O/t -o t --output-format csv -- python3 $R/scripts/single_quantity_probe.py doc egno hyperbolic > $O/probe.json 2> $O/probe.err || { tail -5 $O/probe.err; exit 1; }
cp $(find $O/t -name "*kernel_stats.csv" | head -1) $R/profiles/r${RND}_next_rows_kernel_stats.csv || exit 1
cp $O/probe.json $R/profiles/r${RND}_next_rows.json
cp $R/profiles/r${RND}_next_rows_kernel_stats.csv $R/profiles/r${RND}_next_rows.json $R/gpurun_out/ 2>/dev/null
head -30 $R/profiles/r${RND}_next_rows_kernel_stats.csv
