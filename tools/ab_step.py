"""Same-box, same-process A/B of result-neutral launch parameters over the bench's timed step ("same-box A/B for every claim").

usage: python tools/ab_step.py KEY=a,b[,c] [--steps 4] [--reps 3] [--mode pipelined|serial_step]
Builds the engine ONCE (the bench's own Stepper: 8 clips per step, ragged prefill, batched decode), then alternates gvl_debug_set(KEY, v) over
`reps` rounds of `steps` timed steps per value (one untimed step after every switch).  Prints ms per step per value and round, and the per-family
event times (gvl_prof) of one profiled serial step per value.  A lib-level A/B (two builds) is two runs of this tool under GVL_LIB_PATH."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("spec", nargs="+", help="KEY=v0,v1[,v2]; several specs are switched together (position-wise)")
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--mode", default="pipelined", choices=["pipelined", "serial_step"])
    ap.add_argument("--cps", type=int, default=8)
    ap.add_argument("--prof", action="store_true", help="also one hipEvent-profiled serial step per value (GEMM / attention / other ms per clip)")
    args = ap.parse_args()
    keys, vals = [], []
    for s in args.spec:
        k, v = s.split("=")
        keys.append(k)
        vals.append([int(x) for x in v.split(",")])
    nv = len(vals[0])
    assert all(len(v) == nv for v in vals)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    D = bench.Dev(dev)
    cps = args.cps
    eng, geo = bench.build_engine(dev, max_segs=12 * cps, new_tokens=12, clips_per_step=cps)
    st = bench.Stepper(eng, geo, 0, 1, 12, pool=2 * cps)
    if args.mode == "pipelined":
        st.pipe_start_multi(cps)
        stepfn = st.pipe_step_multi
    else:
        st.cps, st.clip_batch = cps, True

        def stepfn():
            idx = st.window(cps)
            outs = st.step_multi_serial(idx)
            st.last_idx = idx[-1]
            return outs[-1], 0

    def setv(i):
        for k, v in zip(keys, vals):
            eng.debug_set(k, v[i])

    for i in range(nv):                       # warm every variant's kernels / graphs
        setv(i)
        stepfn(); stepfn()
    D.sync()
    res = {i: [] for i in range(nv)}
    for r in range(args.reps):
        order = list(range(nv)) if r % 2 == 0 else list(range(nv - 1, -1, -1))
        for i in order:
            setv(i)
            stepfn()
            D.sync()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                stepfn()
            D.sync()
            res[i].append(1e3 * (time.perf_counter() - t0) / args.steps)
    out = {"mode": args.mode, "steps": args.steps, "reps": args.reps, "variants": []}
    for i in range(nv):
        ms = res[i]
        out["variants"].append({"set": {k: v[i] for k, v in zip(keys, vals)}, "ms_per_step": [round(x, 2) for x in ms], "mean_ms": round(sum(ms) / len(ms), 2),
                                "clips_per_s": round(1e3 * cps * len(ms) / sum(ms), 3)})
    if args.prof:
        st.cps, st.clip_batch = cps, True
        for i in range(nv):
            setv(i)
            idx = st.window(cps)
            st.step_multi_serial(idx)
            D.sync()
            eng.prof_enable(True)
            st.step_multi_serial(st.window(cps))
            D.sync()
            fam = {}
            for name, cat in (("gemm", 0), ("attn", 1), ("gemv", 2), ("decode_attn", 3), ("other", 4)):
                ms, n, work = eng.prof_read(cat)
                fam[name + "_ms_per_clip"] = round(ms / cps, 3)
                fam[name + "_launches"] = int(n)
            eng.prof_enable(False)
            out["variants"][i]["serial_step_families"] = fam
    print(json.dumps(out, indent=1), flush=True)


if __name__ == "__main__":
    main()
