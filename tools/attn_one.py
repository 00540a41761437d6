"""Run the attention kernel alone (target for rocprofv3 --pmc / timing)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _gvl_bootstrap  # noqa
from grounded_video_llm_amd import engine as E

B, S, H, KV, Dr, causal = (int(x) for x in sys.argv[1:7])
iters = int(sys.argv[7]) if len(sys.argv) > 7 else 3
eng = E.Engine(E.TowerGeometry(max_segs=12), "cuda:0", towers=("iv2",))   # iv2 tower only sizes the workspace arena
for kv in [x for x in os.environ.get("GVL_LAB_SET", "").split(",") if x]:
    k, v = kv.split("="); eng.debug_set(k, int(v))
qkv = torch.randn((B * S, (H + 2 * KV) * Dr), device="cuda").to(torch.bfloat16)
for _ in range(2):
    eng.op_attention(qkv, B, S, H, KV, Dr, Dr ** -0.5, causal)
torch.cuda.synchronize()
eng.prof_enable(True)
for _ in range(iters):
    eng.op_attention(qkv, B, S, H, KV, Dr, Dr ** -0.5, causal)
ms, n, work = eng.prof_read(1)
print(os.environ.get("GVL_LAB_SET", ""), f"attention B{B} S{S} H{H}/{KV} D{Dr} causal{causal}: {ms/n*1e3:.1f} us/launch, {work/ms/1e9:.1f} TFLOP/s algorithmic")
