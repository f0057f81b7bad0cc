#!/bin/bash
# Per-workload kernel traces of bench.py's `secondary` entries (BASELINE configs[2], configs[3] and the doc model; the
# default build and, as <workload>_tuned, the profile-guided build):
# each workload alone under `rocprofv3 --kernel-trace --stats` (the program directly after `--`), one CSV per workload
# under profiles/rNN_secondary_<workload>_kernel_stats.csv, and profiles/rNN_secondary.json with the record bench.py
# prints for it (HIP-event ms, points/s, code-object tag) next to the profile's average kernel duration.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
RND=${1:-04}
O=$R/gpurun_out/prof_secondary
rm -rf $O && mkdir -p $O && cd /tmp && export TMPDIR=/tmp
for W in d5 egno doc d5:tuned egno:tuned doc:tuned; do
  F=${W/:/_}
  rocprofv3 --kernel-trace --stats -d $O/$F -o t --output-format csv -- python3 $R/scripts/secondary_probe.py $W > $O/$F.json 2> $O/$F.err || exit 1
  cp $(find $O/$F -name "*kernel_stats.csv" | head -1) $R/profiles/r${RND}_secondary_${F}_kernel_stats.csv || exit 1
done
python3 - "$O" "$R/profiles/r${RND}_secondary.json" <<'PY'
import csv, glob, json, os, sys
o, out = sys.argv[1], sys.argv[2]
table = {"_comment": "bench.py `secondary` workloads, each alone under rocprofv3 --kernel-trace --stats (scripts/profile_secondary.sh): "
         "`bench_record` is what bench.py prints for the workload in that very run (HIP events), `profile` the tile kernel's row of the "
         "committed CSV; one call of the d5 workload is ONE launch of inflx_sweep_tile_complete with 32 parameter rows"}
for w in ("d5", "egno", "doc", "d5_tuned", "egno_tuned", "doc_tuned"):
    rec = json.loads([ln for ln in open(os.path.join(o, w + ".json")).read().splitlines() if ln.startswith("{")][-1])
    if w.endswith("_tuned"):  # the probe ran the profile-guided build only
        rec = dict(rec["profile_guided"], workload=rec["workload"])
    rows = list(csv.DictReader(open(glob.glob(os.path.join(o, w, "**", "*kernel_stats.csv"), recursive=True)[0])))
    tile = [r for r in rows if r["Name"].startswith("inflx_sweep_tile_complete")][0]
    table[w] = {"bench_record": rec, "profile": {"kernel": tile["Name"], "calls": int(tile["Calls"]), "avg_ms": float(tile["AverageNs"]) / 1e6,
                "min_ms": float(tile["MinNs"]) / 1e6, "max_ms": float(tile["MaxNs"]) / 1e6}, "code_object": rec.get("code_object")}
json.dump(table, open(out, "w"), indent=1)
print(json.dumps({w: (table[w]["bench_record"].get("ms"), table[w]["profile"]) for w in table if not w.startswith("_")}, indent=1))
PY
