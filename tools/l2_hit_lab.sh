# LAB: L2 hit rate (TCC_HIT_sum / TCC_MISS_sum) of the shipped GEMM forms at the bench M -- profiles/r06_gemm_l2_hit.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/l2hit; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
: > $OUT/l2hit.txt
while read name M N K mode; do
  rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d $OUT/d_$name -- python $R/tools/gemm_one.py $M $N $K 0 3 $mode > $OUT/$name.log 2>&1
  python - "$(find $OUT/d_$name -name '*.db' | head -1)" $name >> $OUT/l2hit.txt <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name").fetchall()
for k in sorted(set(r[0] for r in rows if "gemm_" in r[0])):
    d = {r[1]: (r[2], r[3]) for r in rows if r[0] == k}
    h, m = d.get("TCC_HIT_sum", (0, 0)), d.get("TCC_MISS_sum", (0, 0))
    print(f"{sys.argv[2]:10s} {k[:60]:60s} n={h[0]} hit {h[1]/max(h[0],1):.4g} miss {m[1]/max(m[0],1):.4g} per launch  hit rate {h[1]/max(h[1]+m[1],1):.3f}")
PY
  rm -rf $OUT/d_$name
done <<SHAPES
iv2.qkv 196704 4224 1408 rs
iv2.proj 196704 1408 1408 bias_gamma_resid_sq
iv2.fc1 196704 6144 1408 rs_bias_gelu
iv2.fc2 196704 1408 6144 bias_gamma_resid_sq
phi.gu 14076 16384 3072 rs_silu
phi.down 14076 3072 8192 resid_sq
SHAPES
cat $OUT/l2hit.txt
