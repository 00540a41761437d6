"""Summarise a rocprofv3 (rocpd sqlite) kernel trace into a --stats style table (name, calls, total, avg, %).

usage: python tools/rocpd_stats.py <results.db> [out.txt] [--between KERNEL_SUBSTRING]
--between: only the dispatches that START after the end of the first and before the start of the last dispatch whose name contains the substring
(bench.py --plain brackets its timed steps with gvl_trace_marker_kernel): the table then holds the timed region only -- no set-up kernels."""
import re
import sqlite3
import sys


def main(path, out=None, between=None):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    where, note = "", ""
    if between:
        marks = cur.execute(f"select start, end from kernels where {name_col} like ? order by start", (f"%{between}%",)).fetchall()
        if len(marks) < 2:
            raise SystemExit(f"--between {between}: {len(marks)} marker dispatches in the trace (need >= 2)")
        t0, t1 = marks[0][1], marks[-1][0]
        where = f"where start >= {t0} and start <= {t1} and {name_col} not like '%{between}%'"
        note = f"# window: between the first and the last of {len(marks)} {between} dispatches = {(t1 - t0) / 1e6:.3f} ms of wall time on the device"
    rows = cur.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels {where} group by {name_col} order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    lines = [f"# rocprofv3 --kernel-trace summary of {path}", f"# total kernel time {tot/1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches"]
    if note:
        lines.append(note)
        aten = sum(r[1] for r in rows if "at::native" in r[0] or "rocclr" in r[0])
        lines.append(f"# dispatches of at::native / rocclr kernels inside the window: {aten}")
    lines.append(f"{'kernel':<70} {'calls':>7} {'total_ms':>10} {'avg_us':>10} {'min_us':>9} {'max_us':>9} {'pct':>6}")
    for n, c, s, a, mn, mx in rows:
        short = re.sub(r"\(.*", "", n)[:70]
        lines.append(f"{short:<70} {c:>7} {s/1e6:>10.3f} {a/1e3:>10.2f} {mn/1e3:>9.2f} {mx/1e3:>9.2f} {100*s/tot:>6.2f}")
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "w").write(txt + "\n")


if __name__ == "__main__":
    args = sys.argv[1:]
    between = None
    if "--between" in args:
        i = args.index("--between")
        between = args[i + 1]
        del args[i:i + 2]
    main(args[0], args[1] if len(args) > 1 else None, between)
