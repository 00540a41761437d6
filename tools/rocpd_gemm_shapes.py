"""Group GEMM dispatches of a rocprofv3 kernel trace by (kernel template, grid) -> count / avg duration."""
import sqlite3, sys, re
cur = sqlite3.connect(sys.argv[1]).cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
gcol = [c for c in cols if "grid" in c.lower()]
print("# cols:", gcol)
q = f"select name, {gcol[0] if gcol else '0'}, count(*), avg(end-start), sum(end-start) from kernels where name like '%gemm%' group by 1,2 order by 5 desc"
for n, g, c, a, s in cur.execute(q).fetchall():
    print(f"{n.split('(')[0][-60:]:<62} grid={g:<9} n={c:<5} avg_us={a/1e3:9.1f} total_ms={s/1e6:8.2f}")
