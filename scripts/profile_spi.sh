#!/bin/bash
# Where do the empty wave slots of the D5 tile kernel come from?  The SPI's resource-allocation counters (one pass of <= 4 counters
# each, --kernel-trace only, the program directly behind `--`), D5 x 32 against EGNO x 32 -- the kernel with ~8 % empty slots between
# workgroups against the one with ~2 % (profiles/r04_experiments.txt section 7).  Usage: profile_spi.sh [case ...]
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_spi
rm -rf $O && mkdir -p $O && cd /tmp && export TMPDIR=/tmp
export INFLX_PROBE_STAMP=$O/cases.json
export INFLX_PROBE_REPEATS=3
CASES=${@:-"d5:4096:32 egno:4096:32"}
P="python3 $R/scripts/isa_mix_probe.py $CASES"
rocprofv3 --pmc SPI_RA_REQ_NO_ALLOC SPI_RA_REQ_NO_ALLOC_CSN SPI_RA_RES_STALL_CSN SPI_CSN_BUSY --kernel-trace -d $O/p1 -o a -- $P > $O/p1.log 2>&1 || { tail -5 $O/p1.log; exit 1; }
echo "pass 1 done"
rocprofv3 --pmc SPI_RA_LDS_CU_FULL_CSN SPI_RA_VGPR_SIMD_FULL_CSN SPI_RA_WAVE_SIMD_FULL_CSN SPI_RA_SGPR_SIMD_FULL_CSN --kernel-trace -d $O/p2 -o a -- $P > $O/p2.log 2>&1 || { tail -5 $O/p2.log; exit 1; }
echo "pass 2 done"
rocprofv3 --pmc SPI_RA_TMP_STALL_CSN SPI_RA_BAR_CU_FULL_CSN SPI_RA_TGLIM_CU_FULL_CSN SPI_RA_WVLIM_STALL_CSN --kernel-trace -d $O/p3 -o a -- $P > $O/p3.log 2>&1 || { tail -5 $O/p3.log; exit 1; }
echo "pass 3 done"
rocprofv3 --pmc SQ_LEVEL_WAVES SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $O/p4 -o a -- $P > $O/p4.log 2>&1 || { tail -5 $O/p4.log; exit 1; }
echo "pass 4 done"
rocprofv3 --pmc SPI_CSN_WAVE SPI_CSN_NUM_THREADGROUPS SPI_CSN_WINDOW_VALID SQ_WAVE_CYCLES --kernel-trace -d $O/p5 -o a -- $P > $O/p5.log 2>&1 || { tail -5 $O/p5.log; exit 1; }
echo "pass 5 done"
cd $R && python3 - $O <<'PY'
import glob, json, os, sys
sys.path.insert(0, "scripts")
import isa_mix_report as r
cases = json.load(open(os.path.join(sys.argv[1], "cases.json")))
out = {c["case"]: {"code_object": c["code_object"]} for c in cases}
for db in sorted(glob.glob(os.path.join(sys.argv[1], "p*", "**", "*.db"), recursive=True)):
    rows = r.dispatches(db)
    k = 0
    for c in cases:
        k += c["repeats"]
        row = rows[k - 1]  # the last repetition of the case
        rec = out[c["case"]]
        for name, v in row.items():
            if name == "us":
                rec.setdefault("us_by_pass", []).append(round(v, 1))
            elif name != "kernel":
                rec[name] = v
json.dump(out, open(os.path.join(sys.argv[1], "spi.json"), "w"), indent=1)
for case, rec in out.items():
    print(case, {a: (round(b, 1) if isinstance(b, float) else b) for a, b in rec.items()})
PY
