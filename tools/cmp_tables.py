"""Compare two per-shape kernel tables written by `bench.py --table-dir` (same box, same call): time per clip and rate of
every (kernel, shape) row, largest absolute differences first.  Usage: python tools/cmp_tables.py a.json b.json [min_ms]"""
import json
import sys

a, b = (json.load(open(p)) for p in sys.argv[1:3])
min_ms = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
ka = {(r["kernel"], r["shape"]): r for r in a["by_shape"]}
kb = {(r["kernel"], r["shape"]): r for r in b["by_shape"]}
rows = []
for k in set(ka) | set(kb):
    ma, mb = ka.get(k, {}).get("ms", 0.0), kb.get(k, {}).get("ms", 0.0)
    rows.append((mb - ma, k, ma, mb))
rows.sort(key=lambda r: -abs(r[0]))
print("clip kernel ms: %.1f -> %.1f" % (a["clip_kernel_ms"], b["clip_kernel_ms"]))
for d, k, ma, mb in rows:
    if abs(d) < min_ms:
        break
    print("%+7.2f ms  %-28s %-42s %8.2f -> %8.2f  (%+.1f %%)" % (d, k[0][:28], k[1][:42], ma, mb, 100 * d / ma if ma else 0))
