#!/bin/bash
# HBM / fabric bytes per launch of the chained passes: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes
# (MI355X_MICROARCH.md, HBM/rocprofv3 section), summarised per kernel name.
#   tools/pmc_traffic.sh <tag> <script> [script args...]        e.g.  tools/pmc_traffic.sh c3_fwd time_forward.py 20 10
# Writes gpurun_out/pmc_<tag>_traffic_raw.json.  Run through gpurun from the repo root.
set -u
TAG=$1; shift
SCRIPT=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_${TAG}_$c -- python3 $R/tools/$SCRIPT "$@" > $R/gpurun_out/pmc_${TAG}_$c.log 2>&1 || echo "$c failed"
done
cd $R; python3 - "$TAG" <<'PY'
import collections, csv, glob, json, statistics, sys
tag = sys.argv[1]
out = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(f"gpurun_out/pmc_{tag}_{c}/**/*counter_collection.csv", recursive=True)
    if not files:
        continue
    by = collections.defaultdict(list)
    for r in csv.DictReader(open(files[0])):
        if r["Counter_Name"] == c and ("k_chain" in r["Kernel_Name"] or "k_factor" in r["Kernel_Name"]):
            by[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    for k, v in by.items():
        out[k][c] = {"dispatches": len(v), "median_KB": statistics.median(v), "mean_KB": statistics.mean(v), "min_KB": min(v), "max_KB": max(v)}
json.dump(out, open(f"gpurun_out/pmc_{tag}_traffic_raw.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
