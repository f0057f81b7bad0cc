#!/bin/bash
# The round's whole evidence pass on a GPU box, in the order that lets every record quote the one before it:
#   1. tile-kernel counters (VALU per point, traffic, LDS conflicts)        -> profiles/rNN_valu.json, rNN_tile_kernels_4096_kernel_stats.csv
#   2. dynamic instruction mix, shader clock, issue-weighted roofline          -> profiles/rNN_isa_mix.json, rNN_isa_mix_kernel_stats.csv
#   3. the secondary workloads alone under rocprofv3 --kernel-trace --stats   -> profiles/rNN_secondary*.{csv,json}
#   4. the bench command under --kernel-trace --stats, the traffic passes, then the plain bench line -> profiles/rNN_bench*.json, rNN_traffic.json
# Everything is copied to gpurun_out/profiles_rNN/ as well (gpurun merges only gpurun_out/ back).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
RND=${1:-04}
cd $R
bash scripts/profile_tile.sh $RND > gpurun_out/profile_tile.log 2>&1 || { echo "profile_tile failed"; tail -5 gpurun_out/profile_tile.log; exit 1; }
echo "tile counters done"
bash scripts/profile_isa_mix.sh $RND > gpurun_out/profile_isa_mix.log 2>&1 || { echo "profile_isa_mix failed"; tail -5 gpurun_out/profile_isa_mix.log; exit 1; }
cp gpurun_out/prof_isa/kernel_stats.csv profiles/r${RND}_isa_mix_kernel_stats.csv
echo "isa mix done"
bash scripts/profile_secondary.sh $RND > gpurun_out/profile_secondary.log 2>&1 || { echo "profile_secondary failed"; tail -5 gpurun_out/profile_secondary.log; exit 1; }
echo "secondary done"
bash scripts/profile_bench.sh $RND > gpurun_out/profile_bench.log 2>&1 || { echo "profile_bench failed"; tail -5 gpurun_out/profile_bench.log; exit 1; }
echo "bench done"
mkdir -p gpurun_out/profiles_r${RND} && cp profiles/r${RND}_* gpurun_out/profiles_r${RND}/
ls gpurun_out/profiles_r${RND}
