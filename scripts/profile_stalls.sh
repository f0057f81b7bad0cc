#!/bin/bash
# Diagnostic counters of the tile kernels (D5 x 32 against EGNO x 32): instruction cache, wave waits, workgroup-launch stalls.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_stalls
rm -rf $O && mkdir -p $O && cd /tmp && export TMPDIR=/tmp
export INFLX_PROBE_STAMP=$O/cases.json
P="python3 $R/scripts/isa_mix_probe.py d5:4096:32 egno:4096:32"
rocprofv3 --pmc SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_INST_ANY --kernel-trace -d $O/p1 -o a -- $P > $O/p1.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_LEVEL_WAVES SQ_CYCLES --kernel-trace -d $O/p2 -o a -- $P > $O/p2.log 2>&1 || exit 1
rocprofv3 --pmc SPI_RA_VGPR_SIMD_FULL_CSN SPI_RA_LDS_CU_FULL_CSN SPI_RA_WAVE_SIMD_FULL_CSN SPI_RA_RES_STALL_CSN --kernel-trace -d $O/p3 -o a -- $P > $O/p3.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace -d $O/p4 -o a -- $P > $O/p4.log 2>&1 || exit 1
cd $R && python3 - $O <<'PY'
import sys, glob, os
sys.path.insert(0, "scripts")
import isa_mix_report as r
out = {}
for db in sorted(glob.glob(os.path.join(sys.argv[1], "p*", "**", "*.db"), recursive=True)):
    rows = r.dispatches(db)
    for name, row in zip(("d5", "d5", "d5", "d5", "egno", "egno", "egno", "egno"), rows):
        out.setdefault(name, {})
        out[name].update({k: v for k, v in row.items() if k != "kernel"})  # the last repetition of each case wins
for k, v in out.items():
    print(k, {a: (round(b, 1) if isinstance(b, float) else b) for a, b in v.items()})
PY
