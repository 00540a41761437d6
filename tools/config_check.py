"""Run one BASELINE.json configuration end to end through the Python mirror with synthetic full-size weights.
  python tools/config_check.py c0|c1|c3|c4   (c0: Phi 8 frames/1 seg; c1: Phi 96f; c3: Llama-3-8B 96f; c4: Llama-3-8B 256f/32 segs)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _gvl_bootstrap  # noqa
from grounded_video_llm_amd import engine as E, prompts as P, synth
from grounded_video_llm_amd.model import LLAVA_NEXT_VIDEO, SyntheticTokenizer

cfg = sys.argv[1]
llm = "phi3.5" if cfg in ("c0", "c1") else "llama3"
frames, segs, new = {"c0": (8, 1, 12), "c1": (96, 12, 12), "c3": (96, 12, 12), "c4": (256, 32, 64)}[cfg]
dev = "cuda:0"
geo = E.TowerGeometry() if llm == "phi3.5" else E.TowerGeometry.llama3_8b()
if llm == "phi3.5":
    geo.rope_short, geo.rope_long = synth.longrope_factors(96)
geo.max_segs, geo.max_seq, geo.max_prefill, geo.kv_pages = segs, 8192, 6656, 128
t0 = time.time()
sd = {"vision_tower": synth.clip_weights(seed="cc.clip", device=dev), "video_encoder": synth.iv2_weights(frames=8, seed="cc.iv2", device=dev),
      "projectors": synth.projector_weights(llm, geo.hidden, seed="cc.proj", device=dev),
      "language_model": synth.llm_weights(geo.kind, geo.hidden, geo.inter, geo.layers, geo.heads, geo.kv_heads, geo.vocab, True, seed="cc.llm", device=dev)}
tok = SyntheticTokenizer(geo.vocab, 300)
model = LLAVA_NEXT_VIDEO(stage="sft", max_txt_len=2048, num_frames=frames, num_segs=segs, llm=llm, geometry=geo, tokenizer=tok, state_dicts=sd, device=dev)
del sd; torch.cuda.empty_cache()
print(f"[{cfg}] model ready in {time.time()-t0:.1f}s; tokens/seg {model.engine.tokens_per_seg}", flush=True)
g = torch.Generator(device=dev); g.manual_seed(42)
samples = {"prompts": [P.build_prompt(llm, "grounding", "Give you a textual query: 'a person opens the door'. When does the described content occur in the video?")],
           "spatial_pixel_values": torch.randn((1, segs, 3, 336, 336), device=dev, generator=g),
           "temporal_pixel_values": torch.randn((1, frames, 3, 224, 224), device=dev, generator=g), "video_ids": ["synthetic"]}
tok.eos_token_id = -1   # never stop early
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = model.generate(samples, do_sample=False, num_beams=1, max_new_tokens=new)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"[{cfg}] generate #{it}: {dt*1e3:.1f} ms -> {1/dt:.2f} clips/s; {len(out[0].split())} tokens; text[:60]={out[0][:60]!r}", flush=True)
print(f"[{cfg}] peak torch mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
