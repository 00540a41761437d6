"""-m gpu: the drop-in boundary exercised WITHOUT Python on the device path: tests/c/e2e_host.c -- plain C99, include/gvl.h + the HIP
runtime's C API -- loads one packed checkpoint file (tools/pack_checkpoint.py's format, written here from the tiny synthetic state dicts), encodes two segments,
splices, prefills and decodes greedily on the GPU.  Its ids must equal the Python host's for the same inputs (same library, same launches: bit-identical) and the
CPU oracle's up to the suite's near-tie rule.  The reference path replaced: models/llava_next_video.py:616-666 (generate)."""
import ctypes as C
import os
import shutil
import subprocess
import tempfile

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import gvl_oracle as O  # noqa: E402
from gpu_util import DEV  # noqa: E402
from grounded_video_llm_amd import engine as E, lib as L, prompts as P, synth, weights as Wt  # noqa: E402
from grounded_video_llm_amd.model import SyntheticTokenizer  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("llm", ["phi3.5", "llama3"])
def test_c_host_runs_the_whole_path_on_the_gpu(llm):
    cc = shutil.which("cc") or shutil.which("gcc")
    if not cc or not os.path.exists("/opt/rocm/include/hip/hip_runtime_api.h"):
        pytest.skip("no C compiler / HIP headers on this box")
    hid, vocab, n_segs, T, max_new = 128, 640, 2, 2, 10
    kind = "phi3" if llm == "phi3.5" else "llama"
    short, long = synth.longrope_factors(32)
    geo = E.TowerGeometry(llm=llm, clip_hidden=64, clip_inter=128, clip_layers=3, clip_heads=4, iv2_dim=64, iv2_inter=128, iv2_depth=3,
                          iv2_heads=4, hidden=hid, inter=256, layers=2, heads=4, kv_heads=4 if kind == "phi3" else 2, vocab=vocab,
                          rope_short=short if kind == "phi3" else None, rope_long=long if kind == "phi3" else None,
                          rope_theta=10000.0 if kind == "phi3" else 500000.0, max_seq=2048, max_segs=6, kv_pages=40, max_prefill=1024, frames_per_seg=T)
    sd = {"vision_tower": synth.clip_weights(64, 128, 3, seed="chost.clip"), "video_encoder": synth.iv2_weights(64, 128, 3, T, seed="chost.iv2"),
          "projectors": synth.projector_weights(llm, hid, 64, 64, seed="chost.proj"),
          "language_model": synth.llm_weights(kind, hid, 256, 2, 4, geo.kv_heads, vocab, True, seed="chost.llm")}
    packed = {}
    packed.update(Wt.pack_clip(sd["vision_tower"], geo.clip_layers - 1))
    packed.update(Wt.pack_iv2(sd["video_encoder"], geo.iv2_depth - 1, T, tokens_per_frame=(geo.iv2_image // geo.iv2_patch) ** 2))
    packed.update(Wt.pack_projectors(sd["projectors"], llm))
    packed.update(Wt.pack_llm(sd["language_model"], kind, geo.layers, geo.heads, geo.kv_heads, geo.max_seq, geo.rope_theta, geo.rope_short, geo.rope_long))
    tok = SyntheticTokenizer(vocab, 300)
    prompt = P.build_prompt(llm, "grounding", "When does the person open the door in the video?")
    ids = O.tokenizer_image_token(prompt, tok, tok.bos_token_id)
    sp = synth.det_tensor("chost.sp", (n_segs, 3, 336, 336))
    tp = synth.det_tensor("chost.tp", (n_segs, 3, T, 224, 224))                # already "(b s) c f h w"
    # ---- the Python host on the same inputs (weights through gvl_load_weight) ----
    eng = E.Engine(geo, DEV)
    try:
        eng.load_packed(packed)
        eng.finalize()
        vis = eng.encode_segments(sp.to(DEV), tp.to(DEV))
        py_ids = eng.generate_ids(eng.splice(ids, vis), max_new, tok.eos_token_id)
        cfg_bytes = bytes(eng.cfg)
    finally:
        eng.close()
    # ---- the C host ----
    with tempfile.TemporaryDirectory() as td:
        Wt.save_packed(os.path.join(td, "w.safetensors"), packed)
        open(os.path.join(td, "cfg.bin"), "wb").write(cfg_bytes)
        sp.numpy().astype(np.float32).tofile(os.path.join(td, "sp.f32"))
        tp.numpy().astype(np.float32).tofile(os.path.join(td, "tp.f32"))
        np.asarray(ids, dtype=np.int64).tofile(os.path.join(td, "ids.i64"))
        exe = os.path.join(td, "e2e_host")
        libdir = os.path.dirname(L.LIB_PATH)
        build = subprocess.run([cc, "-std=c99", "-O1", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"),
                                os.path.join(ROOT, "tests", "c", "e2e_host.c"), "-o", exe, "-L" + libdir, "-l:" + os.path.basename(L.LIB_PATH),
                                "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"], capture_output=True, text=True)
        assert build.returncode == 0, build.stderr
        run = subprocess.run([exe, os.path.join(td, "cfg.bin"), os.path.join(td, "w.safetensors"), os.path.join(td, "sp.f32"), os.path.join(td, "tp.f32"),
                              os.path.join(td, "ids.i64"), str(n_segs), str(len(ids)), str(max_new), str(tok.eos_token_id)], capture_output=True, text=True, timeout=300)
        assert run.returncode == 0, run.stdout + run.stderr
    line = [l for l in run.stdout.splitlines() if "IDS:" in l][0]
    c_ids = [int(x) for x in line.split("IDS:")[1].split()]
    print(f"[parity] C host ({llm}): {line.strip()} ; python host {py_ids}")
    assert int(line.split()[1]) == len(packed)
    assert c_ids == py_ids, "a C host and the Python host drive the same library: the ids must be identical"
    # ---- and the oracle (bf16-emulated, as tests/test_gpu_generate.py) ----
    ref_vis = O.encode_images(sp[None], tp.permute(0, 2, 1, 3, 4).reshape(1, n_segs * T, 3, 224, 224), sd["vision_tower"], sd["video_encoder"], sd["projectors"], llm,
                              clip_layers=3, clip_heads=4, iv2_depth=3, iv2_heads=4, emu=True)[0]
    ocfg = O.LLMConfig(kind, hid, 256, 2, 4, geo.kv_heads, vocab, 1e-5, geo.rope_theta, 131072, 4096, geo.rope_short, geo.rope_long)
    ref_emb = O.splice(torch.tensor(ids), ref_vis, sd["language_model"]["model.embed_tokens.weight"], emu=True)
    ref_ids, margins, scales = O.greedy_generate(ocfg, sd["language_model"], ref_emb, max_new, tok.eos_token_id, emu=True, return_margins=True, return_scales=True)
    for i, (a, b) in enumerate(zip(c_ids, ref_ids)):
        if a != b:
            assert margins[i] < min(2 * 2e-2 * scales[i], 0.25), f"token {i}: {a} vs oracle {b}, margin {margins[i] / scales[i]:.3e} of the logit scale"
            break
