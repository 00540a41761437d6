"""MFMA utilisation of the GEMM kernels of ONE rocprofv3 PMC pass (SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES) over tools/gemm_one.py.
usage: python tools/mfma_util.py <results.db> <label> <M> <N> <K>   -> one line appended to stdout
  SQ_VALU_MFMA_BUSY_CYCLES = matrix-pipe busy cycles summed over the 1024 SIMDs (MI355X_MICROARCH.md: = 32 x N_mfma for 32x32x16 bf16);
  SQ_BUSY_CYCLES = busy cycles summed over the 32 shader engines -> / 32 = kernel duration in shader cycles (the clock under load).
  utilisation = MFMA busy / (1024 x duration cycles);  algorithmic = 2MNK / (duration x 2.5 PFLOP/s dense bf16 peak)."""
import sqlite3
import sys

db, label, M, N, K = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
cur = sqlite3.connect(db).cursor()
rows = cur.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name").fetchall()
mf = sum(r[3] for r in rows if "gemm_" in r[0] and r[1] == "SQ_VALU_MFMA_BUSY_CYCLES")
bz = sum(r[3] for r in rows if "gemm_" in r[0] and r[1] == "SQ_BUSY_CYCLES")
n = max((r[2] for r in rows if "gemm_" in r[0] and r[1] == "SQ_BUSY_CYCLES"), default=0)
kcols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
namec = "name" if "name" in kcols else [c for c in kcols if "name" in c][0]
dur = cur.execute(f"select sum(end-start), count(*) from kernels where {namec} like '%gemm_%'").fetchone()
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 5
if not bz or not dur[0]:
    print(f"{label:18s} no counters"); sys.exit(0)
cyc = bz / 32.0
util = mf / (1024.0 * cyc)
sec = dur[0] * 1e-9
tf = 2.0 * M * N * K * iters / sec / 1e12
print(f"{label:18s} M={M:6d} N={N:6d} K={K:5d}  kernels/launch={dur[1] / iters:4.1f}  {1e6 * sec / iters:8.1f} us  {tf:7.1f} TFLOP/s = {tf / 2500:5.3f} of 2.5 PF  "
      f"clock {cyc / (sec * 1e9) :5.2f} GHz  MFMA pipe busy {100 * util:5.1f} %  (algorithmic MFMA cycles / busy = {2.0 * M * N * K * iters / 1024.0 / mf:5.3f})")
