"""CLI with the flag names of the reference's inference.py (:13-51), running on the MI355X-native path.

    python inference.py --llm phi3.5 --video_path clip.mp4 --ckpt_path ... [--synthetic]

Differences from the reference CLI: boolean flags parse properly, `--device`
must be a HIP device, `--seed` also seeds the device sampler (sampling = temperature -> top-k 50 -> top-p on the device, the
HF semantics of the reference's defaults `--do_sample True --temperature 0.2`; beam search is not built).  `--synthetic`
runs the same plumbing on seeded random weights, synthetic frames and the stand-in tokenizer (there are no
checkpoints, tokenizer files or video decoders in the offline image).
"""
from __future__ import annotations

import argparse
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import _gvl_bootstrap  # noqa: E402,F401
from grounded_video_llm_amd import prompts as P, synth  # noqa: E402
from grounded_video_llm_amd.engine import TowerGeometry  # noqa: E402
from grounded_video_llm_amd.model import LLAVA_NEXT_VIDEO, SyntheticTokenizer  # noqa: E402


def _bool(s: str) -> bool:
    return str(s).lower() in ("1", "true", "yes", "y")


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--device", type=str, default="cuda:0")
    p.add_argument("--model", type=str, default="llava_next_video", choices=["llava_next_video"])
    p.add_argument("--llm", type=str, default="phi3.5", choices=["llama3", "vicuna", "phi3.5"])
    p.add_argument("--stage", type=str, default="sft", choices=["pretrain", "grounded", "sft"])
    p.add_argument("--max_txt_len", type=int, default=2048)
    p.add_argument("--num_temporal_tokens", type=int, default=300)
    p.add_argument("--num_frames", type=int, default=96)
    p.add_argument("--num_segs", type=int, default=12)
    p.add_argument("--lora", type=_bool, default=True)
    p.add_argument("--attn_implementation", type=str, default="flash_attention_2", choices=["eager", "flash_attention_2"])
    p.add_argument("--config_path", type=str, default="weight_path/Phi-3.5-vision-instruct")
    p.add_argument("--tokenizer_path", type=str, default="weight_path/Phi-3.5-mini-instruct")
    p.add_argument("--pretrained_video_path", type=str, default="weight_path/internvideo/vision-encoder-InternVideo2-stage2_1b-224p-f4.pt")
    p.add_argument("--pretrained_vision_proj_llm_path", type=str, default="weight_path/Phi-3.5-vision-instruct-seperated/")
    p.add_argument("--ckpt_path", type=str, default="weight_path/ckpt/sft_llava_next_video_phi3.5_mix_sft_multi_modal_projector_video_projecter_language_model.pth")
    p.add_argument("--prompt_grounding", type=str, default="Give you a textual query: 'The female host wearing purple clothes is reporting news in the studio'. When does the described content occur in the video? Please return the start and end timestamps.")
    p.add_argument("--prompt_videoqa", type=str, default="Question: What does this TV news report about?\nOptions:\n(A) thievery\n(B) community violence incidents\n(C) fashion show\n(D) aging population")
    p.add_argument("--prompt_referring", type=str, default="What is happening from 70 seconds to 80 seconds?")
    p.add_argument("--video_path", type=str, default="./experiments/_3klvlS4W7A.mp4")
    p.add_argument("--do_sample", type=_bool, default=True)
    p.add_argument("--num_beams", type=int, default=1)
    p.add_argument("--max_new_tokens", type=int, default=2048)
    p.add_argument("--temperature", type=float, default=0.2)
    p.add_argument("--top_p", type=float, default=None)
    p.add_argument("--share_visual", type=_bool, default=False, help="encode the video ONCE for the three prompts and batch them (the reference re-encodes per prompt)")
    p.add_argument("--synthetic", action="store_true", help="seeded random weights / frames / tokenizer (offline image)")
    p.add_argument("--synthetic_scale", type=str, default="small", choices=["small", "full"])
    return p.parse_args(argv)


def read_frames(video_path: str, num_frames: int):
    """uint8 [T,3,H,W], fps, vlen, duration.  decord / av are not installed in the offline image: real videos need one of them."""
    try:
        from decord import VideoReader
    except Exception as e:
        raise RuntimeError("video decoding needs `decord` (not in this image); use --synthetic") from e
    vr = VideoReader(video_path, num_threads=1)
    vlen, fps = len(vr), float(vr.get_avg_fps())
    idx = P.sample_frame_indices(num_frames, vlen)
    frames = torch.from_numpy(vr.get_batch(idx).asnumpy()).permute(0, 3, 1, 2)
    return frames, fps, vlen, vlen / fps


def create_prompt(args, mode: str, duration: float) -> str:
    text = {"grounding": args.prompt_grounding, "qa": args.prompt_videoqa, "referring": args.prompt_referring}[mode]
    return P.build_prompt(args.llm, mode, text, duration, args.num_temporal_tokens)


def create_inputs(args, mode: str, frames_u8: torch.Tensor, duration: float, engine):
    """inference.py:65-134 of the reference.  frame_transform (Resize bicubic + CenterCrop + ToTensor + Normalize,
    mm_utils/utils.py:153-183) runs on the GPU, bit-exact to the reference's PIL chain (gvl_preprocess_frames): the uint8
    frames are uploaded once instead of 74 MB of f32 pixels after 108 CPU resizes."""
    fr = frames_u8.to(args.device)
    temporal = engine.preprocess_frames(fr, 224, P.INTERNVIDEO_MEAN, P.INTERNVIDEO_STD).unsqueeze(0)
    sel = P.spatial_indices(args.num_frames, args.num_segs)
    spatial = engine.preprocess_frames(fr[sel], 336, P.OPENAI_DATASET_MEAN, P.OPENAI_DATASET_STD).unsqueeze(0)
    prompt = create_prompt(args, mode, duration)
    return {"video_ids": [args.video_path], "question_ids": [args.video_path], "prompts": [prompt],
            "temporal_pixel_values": temporal.to(args.device), "spatial_pixel_values": spatial.to(args.device)}


def main(argv=None):
    args = parse_args(argv)
    torch.manual_seed(args.seed)
    np.random.seed(args.seed)
    if args.synthetic:
        if args.synthetic_scale == "full":
            geo = {"phi3.5": TowerGeometry, "llama3": TowerGeometry.llama3_8b, "vicuna": TowerGeometry.vicuna_7b}[args.llm]()
        else:
            geo = TowerGeometry(llm=args.llm, clip_layers=3, iv2_depth=3, hidden=512, inter=1024, layers=2, heads=4, kv_heads=4, vocab=2048,
                                max_seq=4608, max_prefill=4096, kv_pages=96)   # 6144 tokens: one 96-frame sequence + max_txt_len 2048
        if geo.kind == "phi3":
            geo.rope_short, geo.rope_long = synth.longrope_factors(geo.hidden // geo.heads)
        geo.max_segs = args.num_segs
        d = args.device
        sd = {"vision_tower": synth.clip_weights(geo.clip_hidden, geo.clip_inter, geo.clip_layers, seed="cli.clip", device=d),
              "video_encoder": synth.iv2_weights(geo.iv2_dim, geo.iv2_inter, geo.iv2_depth, args.num_frames // args.num_segs, seed="cli.iv2", device=d),
              "projectors": synth.projector_weights(args.llm, geo.hidden, seed="cli.proj", device=d),
              "language_model": synth.llm_weights(geo.kind, geo.hidden, geo.inter, geo.layers, geo.heads, geo.kv_heads, geo.vocab, True, seed="cli.llm", device=d)}
        tok = SyntheticTokenizer(geo.vocab, args.num_temporal_tokens)
        model = LLAVA_NEXT_VIDEO(stage=args.stage, max_txt_len=args.max_txt_len, num_frames=args.num_frames, num_segs=args.num_segs,
                                 num_temporal_tokens=args.num_temporal_tokens, lora=args.lora, llm=args.llm, geometry=geo, tokenizer=tok,
                                 state_dicts=sd, device=args.device)
        frames = torch.randint(0, 256, (args.num_frames, 3, 360, 480), dtype=torch.uint8)
        duration = 118.3
        args.max_new_tokens = min(args.max_new_tokens, 16)
    else:
        model = LLAVA_NEXT_VIDEO(stage=args.stage, max_txt_len=args.max_txt_len, num_frames=args.num_frames, num_segs=args.num_segs,
                                 num_temporal_tokens=args.num_temporal_tokens, lora=args.lora, llm=args.llm,
                                 attn_implementation=args.attn_implementation, config_path=args.config_path, tokenizer_path=args.tokenizer_path,
                                 pretrained_video_path=args.pretrained_video_path,
                                 pretrained_vision_proj_llm_path=args.pretrained_vision_proj_llm_path, device=args.device,
                                 ckpt_path=args.ckpt_path)          # base + fine-tuned overlay packed once (inference.py:156-162)
        frames, fps, vlen, duration = read_frames(args.video_path, args.num_frames)

    kw = {"do_sample": args.do_sample, "num_beams": args.num_beams, "max_new_tokens": args.max_new_tokens, "temperature": args.temperature, "top_p": args.top_p, "seed": args.seed}
    outs = {}
    modes = ("grounding", "qa", "referring")
    if args.share_visual:
        # one pre-processing + one vision encode for the three prompts (the reference re-encodes the video per prompt)
        per_mode = [create_inputs(args, mode, frames, duration, model.engine) for mode in modes[:1]]
        prompts = [per_mode[0]["prompts"][0]] + [create_prompt(args, mode, duration) for mode in modes[1:]]
        texts = model.generate_shared(per_mode[0], prompts, **kw)
        outs = {mode: (p, t) for mode, p, t in zip(modes, prompts, texts)}
    else:
        for i, mode in enumerate(modes):
            samples = create_inputs(args, mode, frames, duration, model.engine)
            # one sampler seed per call (HF's generator state advances between the reference's three generate() calls; the same seed
            # for all three would make their draws perfectly correlated)
            outs[mode] = (samples["prompts"][0], model.generate(samples, **{**kw, "seed": args.seed + i})[0])
    print("\n******grounding example******")
    print(outs["grounding"][0])
    print(P.parse_time_interval(outs["grounding"][1], duration, args.num_temporal_tokens, args.llm if args.llm != "vicuna" else "llama3"))
    print("\n******referring example******")
    print(outs["referring"][0])
    print(outs["referring"][1])
    print("\n******videoqa example******")
    print(outs["qa"][0])
    print(outs["qa"][1])


if __name__ == "__main__":
    main()
