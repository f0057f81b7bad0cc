#!/usr/bin/env python3
"""Merge the rocprofv3 --pmc databases of scripts/profile_isa_mix.sh with the static mix of the hot loop (scripts/isa_mix.py).

usage: isa_mix_report.py OUT.json CASES.json DB [DB ...]

Per case (model, grid, parameter rows, build) the record holds
  * the DYNAMIC VALU mix of the whole dispatch: wavefront-instructions per class from SQ_INSTS_VALU_{FMA,MUL,ADD,TRANS}_F64,
    _INT32, _INT64, _CVT and the rest of SQ_INSTS_VALU (comparisons, selects, moves, v_div_scale / v_div_fixup, v_ldexp ...);
  * the effective shader clock of that dispatch, GRBM_GUI_ACTIVE / 8 XCDs / duration (MI355X_MICROARCH.md, DVFS give-back);
  * the issue-weighted VALU time: sum over classes of (wavefront-instructions x issue cost) x 4 cycles / (1024 SIMDs x clock),
    costs from profiles/r02_valu_rates.txt (scripts/isa_mix.py COST) -- the time the kernel would take if its VALU
    instructions issued back to back on every SIMD at the clock the chip actually held;
  * `frac_issue_weighted` = that time / the dispatch's duration in the same pass: the share of the weighted VALU issue
    roofline the kernel reaches;
  * `weighted_cycles_per_point`, the figure bench.py prices its own timing with, and the static mix of the hot loop.
"""
import collections
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import isa_mix  # noqa: E402

SIMDS = 256 * 4
KERNEL = "inflx_sweep_tile_complete"
# dynamic classes the counters separate -> issue cost in v_fma_f64 units (isa_mix.COST)
DYN_COST = {"FMA_F64": isa_mix.COST["fma_f64"], "MUL_F64": isa_mix.COST["mul_f64"], "ADD_F64": isa_mix.COST["add_f64"], "TRANS_F64": isa_mix.COST["trans_f64"],
            "INT32": isa_mix.COST["valu_32"], "INT64": isa_mix.COST["valu_64_int"], "CVT": isa_mix.COST["cvt"]}  # fmt: skip
REST_COST = 1.0  # comparisons 1.03, selects 1.02, v_div_scale / fixup 1.0-1.03, v_ldexp 0.97, v_mov_b64 0.89


def dispatches(db):
    con = sqlite3.connect(db)
    tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
    g = lambda s: [t for t in tabs if t.startswith(s)][0]  # noqa: E731
    q = f"""select d.id, s.kernel_name, d.start, d.end, i.name, sum(e.value) from {g('rocpd_kernel_dispatch')} d
            join {g('rocpd_info_kernel_symbol')} s on d.kernel_id = s.id join {g('rocpd_pmc_event')} e on e.event_id = d.event_id
            join {g('rocpd_info_pmc')} i on e.pmc_id = i.id group by d.id, i.name order by d.id"""
    per = collections.OrderedDict()
    for id_, kernel, start, end, name, value in con.execute(q):
        rec = per.setdefault(id_, {"kernel": kernel.replace(".kd", ""), "us": (end - start) / 1e3})
        rec[name] = value
    return [r for r in per.values() if r["kernel"] == KERNEL]


def main(out_path, cases_path, dbs):
    cases = json.load(open(cases_path))
    passes = [dispatches(db) for db in dbs]
    want = sum(c["repeats"] for c in cases)
    table = {
        "_what": "tile kernels: dynamic VALU instruction mix, effective shader clock and issue-weighted VALU roofline (scripts/profile_isa_mix.sh)",
        "_cost": {"units": "issue time of one wavefront-instruction relative to v_fma_f64 = 4 shader cycles (profiles/r02_valu_rates.txt, 4 waves per SIMD)", "dynamic": DYN_COST, "rest": REST_COST, "static": isa_mix.COST},
        "code_objects": {},
    }
    for rows in passes:
        assert len(rows) == want, (len(rows), want)
    k = 0
    for c in cases:
        k += c["repeats"]
        merged = {}
        durations = {}
        for rows in passes:
            r = rows[k - 1]  # last repetition of the case
            for name, v in r.items():
                if name not in ("kernel", "us"):
                    merged[name] = v
                    durations[name] = r["us"]
        pts = c["P"] * c["n"] * c["n"]
        key = c["model"] + (":tuned" if c["tuned"] else "")
        rec = {"case": c["case"], "kernel": KERNEL, "grid": f"{c['n']}x{c['n']} x {c['P']} parameter rows in one dispatch", "code_object": c["code_object"], "regrouped": c["regrouped"]}
        total = merged.get("SQ_INSTS_VALU")
        dyn = {cls: merged.get(f"SQ_INSTS_VALU_{cls}") for cls in DYN_COST}
        if total is not None and all(v is not None for v in dyn.values()):
            rest = total - sum(dyn.values())
            weighted = sum(dyn[cls] * DYN_COST[cls] for cls in dyn) + rest * REST_COST
            rec["valu_wave_insts"] = total
            rec["valu_insts_per_point"] = total * 64 / pts
            rec["dynamic_mix_per_point"] = {**{cls: v * 64 / pts for cls, v in dyn.items()}, "rest": rest * 64 / pts}
            rec["weighted_fma_units_per_point"] = weighted * 64 / pts
            # cycles one SIMD spends issuing the VALU work of one grid point: 4 cycles per fma unit per wavefront-instruction, 64 points per wavefront
            rec["weighted_cycles_per_point"] = 4.0 * weighted / pts
        if "GRBM_GUI_ACTIVE" in merged:
            us = durations["GRBM_GUI_ACTIVE"]
            rec["clock_GHz"] = merged["GRBM_GUI_ACTIVE"] / 8.0 / (us * 1e3)
            rec["clock_pass_us"] = us
        if "SQ_BUSY_CYCLES" in merged:
            rec["SQ_BUSY_CYCLES"] = merged["SQ_BUSY_CYCLES"]
        if "weighted_cycles_per_point" in rec and "clock_GHz" in rec:
            t_pred_us = rec["weighted_cycles_per_point"] * pts / (SIMDS * rec["clock_GHz"] * 1e3)
            rec["issue_weighted_valu_time_us"] = t_pred_us
            # against the duration of the pass in which the clock was read (same profiling overhead, same DVFS state)
            rec["frac_issue_weighted"] = t_pred_us / rec["clock_pass_us"]
            rec["durations_us_by_pass"] = sorted(set(round(v, 1) for v in durations.values()))
            rec["peak_points_per_s_at_that_clock"] = SIMDS * rec["clock_GHz"] * 1e9 / rec["weighted_cycles_per_point"]
        if "SQ_ACTIVE_INST_VALU" in merged and "SQ_WAVE_CYCLES" in merged:
            rec["valu_active_share_of_wave_cycles"] = merged["SQ_ACTIVE_INST_VALU"] / merged["SQ_WAVE_CYCLES"]
        for name in ("SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_WR", "SQ_WAVES", "GRBM_GUI_ACTIVE"):
            if name in merged:
                rec[name] = merged[name]
        try:
            static = isa_mix.analyse(c["model"], c["tuned"], KERNEL)
            assert static["code_object"] == c["code_object"], (static["code_object"], c["code_object"])
            rec["static_hot_loop"] = {k2: static["hot_loop"][k2] for k2 in ("instructions", "valu", "valu_by_class", "valu_issue_weighted_fma_units", "valu_issue_cycles_per_wave_pass", "non_valu")}
        except Exception as exc:  # noqa: BLE001 -- the dynamic record stands on its own
            rec["static_hot_loop"] = {"error": str(exc)[:200]}
        table[key] = rec
        table["code_objects"][key] = c["code_object"]
    with open(out_path, "w") as fh:
        json.dump(table, fh, indent=1)
    for key, rec in table.items():
        if isinstance(rec, dict) and "case" in rec:
            print(key, {k2: (round(v, 4) if isinstance(v, float) else v) for k2, v in rec.items() if k2 in ("valu_insts_per_point", "weighted_cycles_per_point", "clock_GHz", "issue_weighted_valu_time_us", "clock_pass_us", "frac_issue_weighted", "dynamic_mix_per_point")})


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3:])
