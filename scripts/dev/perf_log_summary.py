"""Sum an ISOCON_PERF_LOG file by call name: calls, wall seconds, kernel ms where the record carries them."""
import json, sys, collections
agg = collections.defaultdict(lambda: [0, 0.0, 0.0, 0])
for ln in open(sys.argv[1]):
    try:
        r = json.loads(ln)
    except ValueError:
        continue
    name = r.get("call") or r.get("name") or "?"
    a = agg[name]
    a[0] += 1
    a[1] += float(r.get("wall_s", r.get("seconds", 0.0)) or 0.0)
    a[2] += float(r.get("kernel_ms", 0.0) or 0.0)
    a[3] += int(r.get("pairs", 0) or 0)
for name, (c, w, k, p) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-50s calls %5d  wall %8.2f s  kernel %9.1f ms  pairs %d" % (name, c, w, k, p))
