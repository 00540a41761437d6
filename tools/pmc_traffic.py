"""HBM-side traffic per kernel family from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) -> profiles/rNN_pmc_traffic.json.

usage: python tools/pmc_traffic.py <fetch.db> <write.db> <out.json> [clips_in_trace]
gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE tallies 64 B per 128-B request for wide coalesced reads, so
read bytes = 2 x FETCH_SIZE; both counters are in KiB-like units of 1 KB?  -- rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KB."""
import json
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _gvl_bootstrap  # noqa: E402,F401
from grounded_video_llm_amd.build import source_sha16  # noqa: E402

# every kernel that bench.py times under GVL_PROF_GEMM / GVL_PROF_ATTN belongs to the family it is divided by (the fused patch embedding is
# timed as GEMM and counted in its launches; InternVideo2's attention runs attn_iv2_pipe_kernel)
FAMILIES = [("gemm", ("gemm_pp_kernel", "gemm_bf16_kernel", "gemm_a4_kernel", "gemm_a4p_kernel", "patch_embed_kernel")), ("attention", ("attn_fwd_kernel", "attn_iv2_pipe_kernel")),
            ("decode_attention", ("decode_attn_kernel",)), ("gemv", ("gemv_kernel", "dgemm_kernel"))]


def per_family(path, counter):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select kernel_name, count(*), sum(value) from counters_collection where counter_name = ? group by kernel_name", (counter,)).fetchall()
    out = {}
    for fam, keys in FAMILIES:
        n = sum(r[1] for r in rows if any(k in r[0] for k in keys))
        v = sum(r[2] for r in rows if any(k in r[0] for k in keys))
        out[fam] = (n, v)
    return out


def main(fetch_db, write_db, out, clips=7):
    f = per_family(fetch_db, "FETCH_SIZE")
    w = per_family(write_db, "WRITE_SIZE")
    fams = {}
    for fam, _ in FAMILIES:
        n = f[fam][0]
        if not n:
            continue
        rd = 2.0 * f[fam][1] * 1024 / n          # KB -> bytes, x2 (gfx950 correction)
        wr = w[fam][1] * 1024 / max(1, w[fam][0])
        fams[fam] = {"launches": n, "read_bytes_per_launch": int(rd), "write_bytes_per_launch": int(wr), "traffic_bytes_per_launch": int(rd + wr),
                     "total_traffic_bytes": int((rd + wr) * n)}
    doc = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python bench.py --plain --mode serial_step --steps 1 --warmup 1 (tools/run_profiles.sh)",
           "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request -> read bytes = 2 x FETCH_SIZE (MI355X_MICROARCH.md, HBM); fabric-side traffic incl. Infinity-Cache hits; counter unit KB",
           "src_sha16": source_sha16(),     # the tree these counters were collected on: bench.py prints `roofline.traffic` only while it still runs that tree
           "clips_in_trace": int(clips),   # bench.py --plain --mode serial_step: (warmup + steps) x 8 clips, nothing else in the process
           "note": "`launches` counts KERNELS (a logical GEMM may run as two kernels after the wave-quantisation split); bench.py divides total_traffic_bytes by clips x logical launches per clip",
           "families": fams}
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps(doc, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:5])
