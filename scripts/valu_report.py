#!/usr/bin/env python3
"""Turn the rocprofv3 --pmc databases of scripts/valu_probe.py into a per-model table:
VALU wavefront-instructions per grid point, share of wave cycles issuing VALU, kernel duration.
usage: valu_report.py OUT.json CODE_OBJECTS.json DB [DB ...]   (one DB per counter group; dispatch order identifies the model;
CODE_OBJECTS.json = the stamps scripts/valu_probe.py wrote: bench.py quotes a record only for the code object it was measured on)"""
import collections
import json
import sqlite3
import sys

CASES = [("hyperbolic", 8192), ("doc", 4096), ("angular", 4096), ("egno", 4096), ("d5", 4096)]
FP64_LANE_RATE = 256 * 4 * 16 * 2.4e9  # lane-instructions/s at full rate (MI355X_MICROARCH.md: 78.6 TFLOP/s FP64 vector = 2 flop x this)


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
    return list(per.values())


def main(out_path, stamp_path, dbs):
    merged = None
    for db in dbs:
        rows = [r for r in dispatches(db) if r["kernel"].startswith("inflx_sweep_") and "rowvals" not in r["kernel"]]
        if merged is None:
            merged = rows
        else:
            assert len(rows) == len(merged), (len(rows), len(merged))
            for a, b in zip(merged, rows):
                assert a["kernel"] == b["kernel"]
                a.update({k: v for k, v in b.items() if k not in ("kernel", "us")})
    # two sweeps per model, in CASES order; keep the second (warm) one
    assert len(merged) == 2 * len(CASES), len(merged)
    table = {"code_objects": json.load(open(stamp_path)), "_units": "WRITE_SIZE / FETCH_SIZE in KB per launch; FETCH_SIZE must be doubled on gfx950 (MI355X_MICROARCH.md); SQ_* cycle counters in quad-cycles"}
    for k, (name, n) in enumerate(CASES):
        r = merged[2 * k + 1]
        pts = n * n
        rec = {"kernel": r["kernel"], "grid": f"{n}x{n}", "kernel_us_under_counters": r["us"]}
        if "SQ_INSTS_VALU" in r:
            rec["valu_wave_insts"] = r["SQ_INSTS_VALU"]
            rec["valu_insts_per_point"] = r["SQ_INSTS_VALU"] * 64 / pts
            rec["valu_issue_ceiling_points_per_s"] = FP64_LANE_RATE / rec["valu_insts_per_point"]
        if "SQ_ACTIVE_INST_VALU" in r and "SQ_WAVE_CYCLES" in r:
            rec["valu_active_share_of_wave_cycles"] = r["SQ_ACTIVE_INST_VALU"] / r["SQ_WAVE_CYCLES"]
        if "WRITE_SIZE" in r and "FETCH_SIZE" in r:
            rec["hbm_bytes_per_launch"] = {"write": r["WRITE_SIZE"] * 1024, "fetch_x2": r["FETCH_SIZE"] * 2048, "algorithmic": 48 * pts}
            rec["traffic_ratio"] = (r["WRITE_SIZE"] * 1024 + r["FETCH_SIZE"] * 2048) / (48 * pts)
        if "SQ_LDS_BANK_CONFLICT" in r and "SQ_LDS_IDX_ACTIVE" in r:
            rec["lds_bank_conflict_share_of_lds_cycles"] = r["SQ_LDS_BANK_CONFLICT"] / max(r["SQ_LDS_IDX_ACTIVE"], 1)
        for c in ("SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_WR", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "WRITE_SIZE", "FETCH_SIZE"):
            if c in r:
                rec[c] = r[c]
        table[name] = rec
    json.dump(table, open(out_path, "w"), indent=1)
    for name, rec in table.items():
        if not isinstance(rec, dict) or "kernel" not in rec:
            continue
        print(name, {k: (round(v, 3) if isinstance(v, float) else v) for k, v in rec.items() if k in ("kernel", "valu_insts_per_point", "valu_active_share_of_wave_cycles", "kernel_us_under_counters", "traffic_ratio", "lds_bank_conflict_share_of_lds_cycles")})


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3:])
