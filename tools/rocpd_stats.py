"""Summarise a rocprofv3 (rocpd sqlite) kernel trace into a --stats style table (name, calls, total, avg, %)."""
import re
import sqlite3
import sys


def main(path, out=None):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by {name_col} order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    lines = [f"# rocprofv3 --kernel-trace summary of {path}", f"# total kernel time {tot/1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches",
             f"{'kernel':<70} {'calls':>7} {'total_ms':>10} {'avg_us':>10} {'min_us':>9} {'max_us':>9} {'pct':>6}"]
    for n, c, s, a, mn, mx in rows:
        short = re.sub(r"\(.*", "", n)[:70]
        lines.append(f"{short:<70} {c:>7} {s/1e6:>10.3f} {a/1e3:>10.2f} {mn/1e3:>9.2f} {mx/1e3:>9.2f} {100*s/tot:>6.2f}")
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "w").write(txt + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
