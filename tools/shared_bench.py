"""inference.py's workload -- three questions about ONE 96-frame video (grounding / QA / referring, inference.py:178-182) -- at full size on
synthetic weights: (a) the reference's flow, one generate() per prompt (vision re-encoded each time), (b) generate_shared without prefix sharing
(one vision encode, three full prefills), (c) generate_shared with the visual prefix prefilled once (gvl_seq_fork + gvl_prefill_extend)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _gvl_bootstrap  # noqa
import torch
from grounded_video_llm_amd import engine as E, synth
from grounded_video_llm_amd.model import LLAVA_NEXT_VIDEO, SyntheticTokenizer

dev = "cuda:0"
geo = E.TowerGeometry(llm="phi3.5", max_seq=4608, max_prefill=4096, kv_pages=0)
geo.rope_short, geo.rope_long = synth.longrope_factors(96)
sd = {"vision_tower": synth.clip_weights(seed="sb.clip", device=dev), "video_encoder": synth.iv2_weights(frames=8, seed="sb.iv2", device=dev),
      "projectors": synth.projector_weights("phi3.5", seed="sb.proj", device=dev), "language_model": synth.llm_weights("phi3", seed="sb.llm", device=dev)}
tok = SyntheticTokenizer(geo.vocab, 300)
model = LLAVA_NEXT_VIDEO(stage="sft", max_txt_len=512, num_frames=96, num_segs=12, llm="phi3.5", geometry=geo, tokenizer=tok, state_dicts=sd, device=dev)
del sd; torch.cuda.empty_cache()
g = torch.Generator(device=dev); g.manual_seed(3)
samples = {"spatial_pixel_values": torch.randn((1, 12, 3, 336, 336), device=dev, generator=g), "temporal_pixel_values": torch.randn((1, 96, 3, 224, 224), device=dev, generator=g), "video_ids": ["v"]}
sys_p = " ".join(f"s{i}" for i in range(34))
prompts = [f"{sys_p} <image> " + " ".join(f"q{j}{i}" for i in range(n)) for j, n in enumerate((40, 25, 60))]
kw = dict(do_sample=False, num_beams=1, max_new_tokens=16)

def timed(fn, n=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, out

t_ref, o_ref = timed(lambda: [model.generate({**samples, "prompts": [p]}, **kw)[0] for p in prompts])
real = model._generate_shared_prefix
model._generate_shared_prefix = lambda rows, vis, mx: None
t_b, o_b = timed(lambda: model.generate_shared(samples, prompts, **kw))
model._generate_shared_prefix = real
t_c, o_c = timed(lambda: model.generate_shared(samples, prompts, **kw))
print(f"[shared] three prompts about one 96-frame video, 16 new tokens each: per-prompt generate() {t_ref:.1f} ms | one vision encode, three full prefills {t_b:.1f} ms | "
      f"visual prefix ({model.last_shared_prefix} tokens) prefilled once {t_c:.1f} ms | answers identical: {o_ref == o_b == o_c}", flush=True)
