"""Offline weight packing: the reference's checkpoint directory -> ONE packed safetensors file for libgvl.

  python tools/pack_checkpoint.py --llm phi3.5 --pretrained_video_path <InternVideo2 .pt> \
      --pretrained_vision_proj_llm_path <Phi-3.5-vision-instruct-seperated/> [--ckpt_path <sft ckpt .pth>] --out weights.gvl.safetensors

Reads exactly the files LLAVA_NEXT_VIDEO.__init__ / inference.py read (models/llava_next_video.py:117-151, inference.py:156-162),
overlays the fine-tuned groups, merges LoRA (W + 2.B.A), fuses q/k/v and gate/up, pads K, interpolates the InternVideo2 temporal
position embedding 4 -> frames_per_seg, builds the RoPE tables, and writes the tensors under the names gvl_load_weight expects.
Engine.load_packed(weights.load_packed_file(path)) then starts without touching the original checkpoints."""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _gvl_bootstrap  # noqa
from grounded_video_llm_amd import weights as Wt
from grounded_video_llm_amd.engine import TowerGeometry
from grounded_video_llm_amd.model import fit_geometry, geometry_from_checkpoint_dirs, load_reference_checkpoints


def pack_all(sd, geo: TowerGeometry, llm: str, stage: str = "sft"):
    packed = {}
    lm = sd["language_model"]
    if stage in ("grounded", "sft"):                     # base weights only: grow the vocabulary like reset_embeddings (:231-268)
        ek = next(k for k in lm if k.endswith("embed_tokens.weight"))
        if geo.vocab > lm[ek].shape[0]:
            lm = Wt.reset_embeddings(lm, geo.vocab - lm[ek].shape[0], geo.lm_head_bias)
    if geo.kind == "phi3" and geo.rope_short is None and geo.rope_orig_max_pos > 0:
        raise ValueError("Phi-3.5: LongRoPE factors missing (config.json rope_scaling not found); refusing to pack plain-RoPE tables")
    packed.update(Wt.pack_clip(sd["vision_tower"], geo.clip_layers - 1))
    # the released InternVideo2 checkpoint is `-f4`: pos_embed holds 4 temporal positions and is interpolated to frames_per_seg
    packed.update(Wt.pack_iv2(sd["video_encoder"], geo.iv2_depth - 1, geo.frames_per_seg, tokens_per_frame=(geo.iv2_image // geo.iv2_patch) ** 2))
    packed.update(Wt.pack_projectors(sd["projectors"], llm))
    packed.update(Wt.pack_llm(lm, geo.kind, geo.layers, geo.heads, geo.kv_heads, geo.max_seq, geo.rope_theta,
                              geo.rope_short, geo.rope_long, geo.rope_max_pos, geo.rope_orig_max_pos))
    return packed


def main(argv=None, geometry=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--llm", default="phi3.5", choices=["phi3.5", "llama3", "vicuna"])
    ap.add_argument("--stage", default="sft", choices=["pretrain", "grounded", "sft"])
    ap.add_argument("--config_path", default=None, help="directory with the HF config.json (rope_scaling factors); default: language_model_seperated/")
    ap.add_argument("--pretrained_video_path", required=True)
    ap.add_argument("--pretrained_vision_proj_llm_path", required=True)
    ap.add_argument("--ckpt_path", default=None)
    ap.add_argument("--num_frames", type=int, default=96)
    ap.add_argument("--num_segs", type=int, default=12)
    ap.add_argument("--num_temporal_tokens", type=int, default=300)
    ap.add_argument("--max_txt_len", type=int, default=2048, help="as given to LLAVA_NEXT_VIDEO at load time: sizes the RoPE tables (max_seq)")
    ap.add_argument("--out", required=True)
    a = ap.parse_args(argv)
    geo = geometry or geometry_from_checkpoint_dirs(a.llm, a.config_path, a.pretrained_vision_proj_llm_path, a.stage, a.num_temporal_tokens)
    geo = fit_geometry(geo, a.llm, a.num_frames, a.num_segs, a.max_txt_len)      # the limits the model constructor will derive
    sd = load_reference_checkpoints(a.llm, a.pretrained_video_path, a.pretrained_vision_proj_llm_path)
    if a.ckpt_path:
        ck = torch.load(a.ckpt_path, map_location="cpu")
        ck = ck.get("model", ck)
        for grp in ("multi_modal_projector", "video_projecter"):
            for k, v in ck.get(grp, {}).items():
                sd["projectors"][f"{grp}.{k}"] = v
        if "language_model" in ck:
            sd["language_model"] = ck["language_model"]
    packed = pack_all(sd, geo, a.llm, a.stage)
    Wt.save_packed(a.out, packed, {"llm": a.llm, "frames_per_seg": str(geo.frames_per_seg), "max_seq": str(geo.max_seq)})
    print(f"wrote {a.out}: {len(packed)} tensors, {sum(v.numel() * v.element_size() for v in packed.values()) / 2**20:.1f} MiB")
    return packed


if __name__ == "__main__":
    main()
