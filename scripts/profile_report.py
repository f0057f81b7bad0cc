#!/usr/bin/env python3
"""Summarise the outputs of scripts/profile_bench.sh into the files kept under profiles/:
kernel stats (name, calls, average ns) and the per-launch HBM traffic of the bench kernels from the
WRITE_SIZE / FETCH_SIZE passes (counter unit KB; FETCH_SIZE doubled on gfx950 as MI355X_MICROARCH.md's HBM
section prescribes).  usage: profile_report.py gpurun_out/prof_bench [round]"""
import csv
import glob
import json
import os
import sys


def find(d, pattern):
    hits = sorted(glob.glob(os.path.join(d, "**", pattern), recursive=True))
    if not hits:
        raise SystemExit(f"no {pattern} under {d}")
    return hits[0]


def kernel_stats(d):
    rows = list(csv.DictReader(open(find(d, "*kernel_stats.csv"))))
    return [{"kernel": r["Name"], "calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "total_ns": float(r["TotalDurationNs"]), "pct": float(r["Percentage"])} for r in rows]


def counter(d, name):
    per = {}
    for r in csv.DictReader(open(find(d, "*counter_collection.csv"))):
        if r["Counter_Name"] != name:
            continue
        per.setdefault(r["Kernel_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
        per[r["Kernel_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return {k: {"launches": len(v), "mean_KB": sum(v.values()) / len(v), "min_KB": min(v.values()), "max_KB": max(v.values())} for k, v in per.items()}


def main(d, rnd="02"):
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(d))), "profiles")
    stats = kernel_stats(os.path.join(d, "stats"))
    with open(os.path.join(out, f"r{rnd}_bench_hyperbolic8192_kernel_stats.csv"), "w") as fh:
        fh.write("Name,Calls,AverageNs,TotalDurationNs,Percentage\n")
        for r in stats:
            fh.write(f"{r['kernel']},{r['calls']},{r['avg_ns']:.1f},{r['total_ns']:.0f},{r['pct']:.2f}\n")
    write, fetch = counter(os.path.join(d, "pmc_write"), "WRITE_SIZE"), counter(os.path.join(d, "pmc_fetch"), "FETCH_SIZE")
    traffic = {
        "_comment": "HBM traffic per launch of the bench kernels: rocprofv3 --pmc WRITE_SIZE and --pmc FETCH_SIZE in separate passes of "
        "`python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline` (scripts/profile_bench.sh). Counter unit KB; bytes = KB*1024; FETCH_SIZE "
        "additionally doubled for gfx950 (MI355X_MICROARCH.md, HBM section).",
        "round": int(rnd),
        "workload": "hyperbolic 8192x8192 complete_analysis AoS",
        "counters": {},
        "bytes_per_launch": {},
    }
    for k in sorted(set(write) | set(fetch)):
        if not k.startswith("inflx_"):
            continue
        w, f = write.get(k), fetch.get(k)
        traffic["counters"][k] = {"write": w, "fetch": f}
        wb = w["mean_KB"] * 1024 if w else 0.0
        rb = f["mean_KB"] * 1024 * 2 if f else 0.0
        traffic["bytes_per_launch"][k] = {"write": wb, "fetch_x2": rb, "total": wb + rb}
        # the record bench.py reads (roofline.traffic); algorithmic bytes: 48 B per grid point of the 8192^2 sweep
        rec = {"write_bytes": wb, "read_bytes_corrected": rb, "traffic_bytes": wb + rb}
        if k == "inflx_sweep_rowstream6":
            rec["algorithmic_bytes"] = 48 * 8192 * 8192
            rec["ratio"] = (wb + rb) / rec["algorithmic_bytes"]
        traffic[k] = rec
    # stamp: the code object these launches came from (bench.py quotes `traffic` only while it loads the same one)
    stamp = None
    for name in ("bench_under_rocprof.json", "bench.json"):
        try:
            lines = [ln for ln in open(os.path.join(d, name)).read().splitlines() if ln.startswith("{")]
            stamp = json.loads(lines[-1])["roofline"]["code_object"]
            break
        except (OSError, IndexError, KeyError, ValueError):
            continue
    traffic["code_objects"] = {k: stamp for k in traffic["bytes_per_launch"]}
    json.dump(traffic, open(os.path.join(out, f"r{rnd}_traffic.json"), "w"), indent=1)
    for name, target in (("bench.json", f"r{rnd}_bench.json"), ("bench_under_rocprof.json", f"r{rnd}_bench_under_rocprof.json")):
        src = os.path.join(d, name)
        if os.path.exists(src):
            lines = [ln for ln in open(src).read().splitlines() if ln.startswith("{")]
            if lines:
                open(os.path.join(out, target), "w").write(lines[-1] + "\n")
    print(json.dumps({"stats": stats[:4], "bytes_per_launch": traffic["bytes_per_launch"]}, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:3])
