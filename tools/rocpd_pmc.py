"""Per-kernel average PMC counter values from a rocprofv3 rocpd sqlite file."""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
filt = sys.argv[2] if len(sys.argv) > 2 else "gemm"
cols = [r[1] for r in cur.execute("pragma table_info(pmc_events)")]
try:
    rows = cur.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection group by kernel_name, counter_name").fetchall()
except Exception as e:
    print("counters_collection query failed:", e)
    ccols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    print("# counters_collection columns:", ccols)
    rows = []
for n, c, cnt, avg, tot in rows:
    if filt in n:
        short = n.split("(")[0][:60]
        print(f"{short:<60} {c:<32} n={cnt:<5} avg={avg:.6g} dur_us={tot/1e3:.1f}")
