#!/usr/bin/env python3
"""(CPU, no GPU needed) Registers, spills, LDS and the executed VALU instruction count of the row loop of
inflx_sweep_tile_complete for an example model under given compiler options -- what decides the speed of the
FP64-VALU-bound tile kernels.
usage: kernel_resources.py MODEL[:opt=val,...] ...   e.g.  d5 d5:hoist_reciprocals=1 egno:hoist_reciprocals=1,waves=1"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import workloads  # noqa: E402
from workloads import example_models  # noqa: E402
from inflatox_amd.compiler import Compiler  # noqa: E402

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
# issue cost relative to v_fma_f64 (scripts/micro/valu_rates.hip, 4 waves per SIMD)
WEIGHT = {"v_rcp_f64": 3.16, "v_rsq_f64": 3.15, "v_sqrt_f64": 3.16, "v_trig_preop_f64": 3.2}


def loops_of(asm: str):
    """The loops of a kernel (target of a backward branch .. that branch), longest first; nested spans are dropped."""
    lines = asm.splitlines()
    addr = {}
    for k, ln in enumerate(lines):
        m = re.search(r"//\s*([0-9A-F]{12}):", ln)
        if m:
            addr[int(m.group(1), 16)] = k
    first = min(addr)
    spans = []
    for k, ln in enumerate(lines):
        m = re.search(r"s_c?branch\w*\s+\d+\s.*<[^>]*\+0x([0-9a-f]+)>", ln)
        here = re.search(r"//\s*([0-9A-F]{12}):", ln)
        if m and here:
            target = first + int(m.group(1), 16)
            if target < int(here.group(1), 16) and target in addr:
                spans.append((addr[target], k))
    spans.sort(key=lambda s: s[0] - s[1])
    keep = []
    for a, b in spans:
        if not any(a >= c and b <= d for c, d in keep):
            keep.append((a, b))
    return [lines[a : b + 1] for a, b in sorted(keep)]


def report(spec_text: str):
    name, _, opt_text = spec_text.partition(":")
    opts = dict(o.split("=") for o in opt_text.split(",") if o)
    spec = example_models.get(name)
    kw = dict(spec.compiler_kwargs)
    flags = list(Compiler.default_hipcc_flags)
    for k, v in opts.items():
        if k == "waves":
            flags.append(f"-DINFLX_MIN_WAVES={v}")
        elif k == "tile_rows":
            flags.append(f"-DINFLX_TILE_ROWS={v}")
        elif k.startswith("D"):
            flags.append(f"-{k}={v}")
        else:
            kw[k] = bool(int(v)) if v.lstrip('-').isdigit() else v  # (hoist_reciprocals=inline)
    art = Compiler(workloads.model_for(name), silent=True, compiler_flags=flags, **kw).compile()
    path = art.shared_object_path
    notes = subprocess.run([READELF, "--notes", path], capture_output=True, text=True).stdout
    blocks = notes.split("- .agpr_count:")
    info = {}
    for b in blocks:
        m = re.search(r"\.name:\s+(\S+)", b)
        if m and m.group(1) == "inflx_sweep_tile_complete":
            for key in ("vgpr_count", "vgpr_spill_count", "sgpr_count", "sgpr_spill_count", "group_segment_fixed_size", "private_segment_fixed_size"):
                mm = re.search(rf"\.{key}:\s+(\d+)", b)
                info[key] = int(mm.group(1)) if mm else None
            mm = re.search(r"^\s*(\d+)\s*$", b.splitlines()[0]) if b.splitlines() else None
            info["agpr_count"] = int(mm.group(1)) if mm else None
    asm = subprocess.run([OBJDUMP, "-d", path], capture_output=True, text=True).stdout
    start = asm.index("<inflx_sweep_tile_complete>:")
    end = asm.index("<inflx_sweep_rows_complete>:")
    hdr = open(art.header_path).read()
    print(
        f"{spec_text:40s} vgpr={info.get('vgpr_count')} spill={info.get('vgpr_spill_count')} scratch={info.get('private_segment_fixed_size')}B lds={info.get('group_segment_fixed_size')}B"
        f" | hoisted quotients: {hdr.count('INFLX_DIVH(') // 2 if 'INFLX_DIVH(' in hdr else 0}, inline: {art.stage_info.get('inline_quotients', 0)} | NU/NR/NC={art.stage_info['nu']}/{art.stage_info['nr']}/{art.stage_info['nc']}",
        flush=True,
    )
    for loop in loops_of(asm[start:end]):
        ops = [ln.split()[0] for ln in loop if ln.strip() and not ln.strip().endswith(":")]
        valu = [o for o in ops if o.startswith("v_")]
        if len(valu) < 100:
            continue
        cost = sum(next((w for k, w in WEIGHT.items() if o.startswith(k)), 1.0) for o in valu)
        print(
            f"    loop: {len(valu)} VALU ({cost:.0f} fma-eq), {sum(o.startswith('v_rcp_f64') for o in valu)} rcp, {sum(o.startswith('v_div_fixup') for o in valu)} ieee-div, "
            f"{sum(o.startswith('ds_') for o in ops)} ds, {sum(o.startswith('scratch_') or o.startswith('buffer_') for o in ops)} scratch",
            flush=True,
        )


if __name__ == "__main__":
    for s in sys.argv[1:] or ["doc", "angular", "egno", "d5"]:
        report(s)
