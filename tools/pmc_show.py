import sqlite3, glob, sys
d, filt = sys.argv[1], sys.argv[2]
f = (glob.glob(d + "/*/*.db") + glob.glob(d + "/*.db"))[0]
cur = sqlite3.connect(f).cursor()
rows = cur.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection group by kernel_name, counter_name").fetchall()
for k, c, n, v, dur in rows:
    if filt in k:
        print(f"{k.split('(')[0][-48:]:<48} {c:<28} n={n:<4} avg={v:.5g} dur_us={dur/1e3:.1f}")
