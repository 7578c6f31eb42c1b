#!/bin/bash
# Build container: regenerate profiles/<TAG>_* and profiles/counters.json for the CURRENT kernel sources (two gpurun calls: the profiled
# passes, then -- with the fresh counters in place -- the bench line that carries them).  Usage: bash scripts/refresh_profile.sh r03i
set -e
TAG=${1:?tag}
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build()"
/usr/local/graft/bin/gpurun --timeout 1100 -- "bash scripts/profile_bench.sh $TAG > gpurun_out/profile_$TAG.log 2>&1; tail -c 300 gpurun_out/profile_$TAG.log"
python scripts/summarize_profile.py $TAG | tail -3
/usr/local/graft/bin/gpurun --timeout 600 -- "python bench.py --steps 5 --warmup 1 > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; tail -c 200 gpurun_out/bench_$TAG.json"
cp gpurun_out/bench_$TAG.json profiles/${TAG}_bench.json
python - <<PY
import json
d = json.load(open("profiles/${TAG}_bench.json"))
r = d["roofline"]
print("ms_per_step %.2f  frac %s  issue-slot frac %s  bound-pass MFMA frac %.3f  counters: %s" % (d["ms_per_step"], r["frac"], r["issue_slots"]["frac"], r["bound_pass"]["frac"], r["counters"]))
PY
