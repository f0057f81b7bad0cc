#!/usr/bin/env python3
"""Static instruction mix of the tile kernels' hot row loop, from the disassembly of a model's code object.

usage: isa_mix.py [--tuned] [--kernel NAME] [--json OUT] MODEL [MODEL ...]

For every model: disassemble `inflx_sweep_tile_complete` (llvm-objdump -d), find its loops (a backward branch and the
range it spans), take the first loop of at least 200 instructions as the HOT row loop (quick point stage + quick epilogue)
and the second as the IEEE redo loop, and count the instructions of each per class.  Every class carries the issue cost
measured on MI355X with scripts/micro/valu_rates.hip (profiles/r02_valu_rates.txt, 4 wavefronts per SIMD): time of one
wavefront-instruction relative to v_fma_f64, whose own issue time is 4 cycles of the real shader clock (64 lanes over 16
FP64 lanes).  The weighted sum is the number of VALU issue cycles one pass through the loop costs a wavefront, i.e. per 64
grid points, with every block of the loop taken once (wave-uniform branches skip some of them at run time: the dynamic
counts come from the SQ_INSTS_VALU_* counters, scripts/profile_isa_mix.sh / isa_mix_report.py).
"""

from __future__ import annotations

import argparse
import collections
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"

# issue cost of one wavefront-instruction in units of v_fma_f64 (= 4 shader cycles), profiles/r02_valu_rates.txt, the
# 4-waves-per-SIMD table (the tile kernels run 3); classes the microbenchmark did not time take the cost of their kin
COST = {
    "fma_f64": 1.00,  # v_fma_f64, v_fmac_f64
    "mul_f64": 1.00,
    "add_f64": 0.96,  # v_add_f64, v_max/min_f64
    "trans_f64": 3.09,  # v_rcp_f64, v_rsq_f64, v_sqrt_f64: quarter-rate pipe
    "div_helper_f64": 1.06,  # v_div_scale_f64 1.03, v_div_fmas_f64 1.16, v_div_fixup_f64 1.00
    "cmp_f64": 1.03,  # v_cmp_*_f64, v_cmp_class_f64
    "other_f64": 0.97,  # v_ldexp_f64 0.97, v_frexp_* 0.94, v_fract 0.93, v_rndne 0.89, v_trig_preop 3.13 (counted apart)
    "trig_preop_f64": 3.13,
    "cvt": 1.00,  # v_cvt_f64_u32 and friends (not timed: one f64 result per lane)
    "mov_b64": 0.89,
    "cndmask": 1.02,  # v_cndmask_b32 with an SGPR / VCC mask held over several selects
    "valu_32": 0.70,  # 32-bit integer / logic / move: 0.65-0.73
    "valu_64_int": 1.00,  # v_lshl_add_u64, v_mad_u64_u32 ...
    "readlane": 0.70,
}


def classify(mnemonic: str) -> str:
    m = mnemonic
    if m.startswith("s_"):
        if m.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_sleep", "s_setprio")):
            return "wait_nop"
        if m.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_swappc")):
            return "branch"
        if m.startswith(("s_load", "s_buffer_load", "s_store", "s_dcache", "s_memtime")):
            return "smem"
        return "salu"
    if m.startswith("ds_"):
        return "lds"
    if m.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "vmem_store" if "store" in m or "atomic" in m else "vmem_load"
    if not m.startswith("v_"):
        return "other"
    if m.startswith(("v_fma_f64", "v_fmac_f64")):
        return "fma_f64"
    if m.startswith("v_mul_f64"):
        return "mul_f64"
    if m.startswith(("v_add_f64", "v_max_f64", "v_min_f64")):
        return "add_f64"
    if m.startswith(("v_rcp_f64", "v_rsq_f64", "v_sqrt_f64")):
        return "trans_f64"
    if m.startswith(("v_div_scale_f64", "v_div_fmas_f64", "v_div_fixup_f64")):
        return "div_helper_f64"
    if m.startswith("v_trig_preop_f64"):
        return "trig_preop_f64"
    if m.startswith("v_cmp") and "f64" in m:
        return "cmp_f64"
    if m.startswith(("v_ldexp_f64", "v_frexp", "v_fract_f64", "v_rndne_f64", "v_floor_f64", "v_ceil_f64", "v_trunc_f64")):
        return "other_f64"
    if m.startswith("v_cvt"):
        return "cvt"
    if m.startswith(("v_mov_b64", "v_pk_mov_b32")):
        return "mov_b64"
    if m.startswith("v_cndmask"):
        return "cndmask"
    if m.startswith(("v_readlane", "v_readfirstlane", "v_writelane", "v_accvgpr")):
        return "readlane"
    if "u64" in m or "i64" in m or "b64" in m:
        return "valu_64_int"
    return "valu_32"


VALU_CLASSES = tuple(COST)
INSN = re.compile(r"^\s+([a-z_0-9]+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):")


def disassemble(path: str, kernel: str):
    text = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", path], capture_output=True, text=True, check=True).stdout
    lines = text.splitlines()
    start = next(i for i, ln in enumerate(lines) if ln.endswith(f"<{kernel}>:"))
    insns = []
    for ln in lines[start + 1 :]:
        if re.match(r"^[0-9a-f]+ <", ln):
            break
        m = INSN.match(ln)
        if m:
            insns.append((int(m.group(3), 16), m.group(1), m.group(2)))
    return insns


def branch_target(addr: int, operand: str) -> int | None:
    try:
        off = int(operand.split()[0])
    except (ValueError, IndexError):
        return None
    if off >= 0x8000:
        off -= 0x10000
    return addr + 4 + 4 * off


def loops(insns):
    """(first index, last index) of every backward branch's range, outermost first by start address, nested ones dropped."""
    index = {a: i for i, (a, _, _) in enumerate(insns)}
    found = []
    for i, (addr, mn, ops) in enumerate(insns):
        if mn.startswith(("s_cbranch", "s_branch")):
            t = branch_target(addr, ops)
            if t is not None and t <= addr and t in index:
                found.append((index[t], i))
    found.sort(key=lambda r: (r[0], -r[1]))
    merged = []
    for a, b in found:
        if merged and a <= merged[-1][1]:  # overlapping / nested: one region
            merged[-1] = (merged[-1][0], max(merged[-1][1], b))
        else:
            merged.append((a, b))
    return merged


def mix(insns, lo, hi):
    c = collections.Counter(classify(mn) for _, mn, _ in insns[lo : hi + 1])
    valu = {k: c.get(k, 0) for k in VALU_CLASSES if c.get(k, 0)}
    weighted = sum(n * COST[k] for k, n in valu.items())
    return {
        "instructions": hi - lo + 1,
        "valu": sum(valu.values()),
        "valu_by_class": valu,
        "valu_issue_weighted_fma_units": weighted,
        "valu_issue_cycles_per_wave_pass": 4.0 * weighted,
        "non_valu": {k: n for k, n in c.items() if k not in VALU_CLASSES},
        "mnemonics": dict(collections.Counter(mn for _, mn, _ in insns[lo : hi + 1] if mn.startswith("v_")).most_common(40)),
    }


def analyse(model: str, tuned: bool, kernel: str):
    import workloads

    _, art = workloads.artifact_for(model, tuned=tuned)
    insns = disassemble(art.shared_object_path, kernel)
    big = [r for r in loops(insns) if r[1] - r[0] + 1 >= 200]
    rec = {
        "model": model,
        "build": "profile-guided" if tuned else "default",
        "kernel": kernel,
        "code_object": os.path.splitext(os.path.basename(art.header_path))[0],
        "kernel_instructions": len(insns),
        "loops_of_200_or_more_instructions": [[hex(insns[a][0]), hex(insns[b][0]), b - a + 1] for a, b in big],
    }
    if big:
        rec["hot_loop"] = mix(insns, *big[0])
        if len(big) > 1:
            rec["ieee_redo_loop"] = mix(insns, *big[1])
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("models", nargs="+")
    ap.add_argument("--tuned", action="store_true")
    ap.add_argument("--kernel", default="inflx_sweep_tile_complete")
    ap.add_argument("--json")
    opt = ap.parse_args()
    out = {"cost_units": "issue time of one wavefront-instruction relative to v_fma_f64 (= 4 shader cycles: 64 lanes over 16 FP64 lanes per SIMD); profiles/r02_valu_rates.txt, 4 waves per SIMD", "cost": COST}
    for name in opt.models:
        rec = analyse(name, opt.tuned, opt.kernel)
        out[name + (":tuned" if opt.tuned else "")] = rec
        hot = rec.get("hot_loop", {})
        print(f"{name}{' (tuned)' if opt.tuned else ''}: kernel {rec['kernel_instructions']} instructions, loops {rec['loops_of_200_or_more_instructions']}")
        if hot:
            print(f"  hot loop: {hot['instructions']} instructions, {hot['valu']} VALU, weighted {hot['valu_issue_weighted_fma_units']:.1f} fma units = {hot['valu_issue_cycles_per_wave_pass']:.0f} issue cycles per 64 points")
            print("   ", hot["valu_by_class"])
            print("    non-VALU:", hot["non_valu"])
    if opt.json:
        with open(opt.json, "w") as fh:
            json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
