"""Median duration per kernel out of a rocprofv3 rocpd database (python tools/dev_rocpd_stats.py results.db [substring ...])."""
import sqlite3, statistics, sys, collections
c = sqlite3.connect(sys.argv[1])
names = {r[0]: r[1] for r in c.execute("select id, kernel_name from rocpd_info_kernel_symbol")}
agg = collections.defaultdict(list)
for k, s, e in c.execute("select kernel_id, start, end from rocpd_kernel_dispatch"):
    agg[names[k]].append(e - s)
for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    if len(sys.argv) > 2 and not any(a in n for a in sys.argv[2:]):
        continue
    v.sort()
    print(f"{n[:90]:90s} n={len(v):5d} med {statistics.median(v) / 1000:8.1f} us  p90 {v[int(0.9 * (len(v) - 1))] / 1000:8.1f} us")
