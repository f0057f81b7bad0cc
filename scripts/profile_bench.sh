#!/bin/bash
# Reproduce the committed profiles of the bench workload on a GPU box (run from the repo root through gpurun):
#   2. rocprofv3 --kernel-trace --stats         -> gpurun_out/prof_bench/stats/*_kernel_stats.csv (+ bench line under rocprof)
#   3. rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE   -> gpurun_out/prof_bench/pmc_{write,fetch}/ (separate passes, kernel trace only)
#   4. plain bench line (last, so that it quotes 3) -> gpurun_out/prof_bench/bench.json
# then scripts/profile_report.py turns them into profiles/rNN_*.  The program itself follows `--` (no env/bash hop).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_bench
mkdir -p $O && cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/stats -o bench --output-format csv -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_under_rocprof.json 2> $O/stats.err || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write -o w --output-format csv -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> $O/pmc_write.err || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch -o f --output-format csv -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2> $O/pmc_fetch.err || exit 1
rm -f $O/bench.json
python3 $R/scripts/profile_report.py $O ${1:-03} > /dev/null || exit 1
# the plain bench line LAST: it quotes the traffic record the passes above have just written for this code object
python3 $R/bench.py --steps 50 --warmup 5 > $O/bench.json 2> $O/bench.err || exit 1
python3 $R/scripts/profile_report.py $O ${1:-03}
