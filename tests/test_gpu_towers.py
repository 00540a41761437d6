"""-m gpu: the vision towers + glue through the C ABI vs (a) the committed goldens produced by the
reference's own modules and (b) the CPU oracle in bf16-emulation mode on the same seeded inputs.
Tolerances (of the output scale): the HIP path computes in bf16 with fp32 accumulation (as the reference GPU path does); every
bound below is <= 1.5 x the error observed on MI355X in round 2 (profiles/r02_parity_observed.txt) -- the kernels are
deterministic, so the margin only has to absorb future re-orderings of fp32 sums, not run-to-run noise.  The full-depth
stacks (23 / 39 layers) are pinned in tests/test_gpu_c0.py."""
import pytest
import torch

pytestmark = pytest.mark.gpu

import gvl_oracle as O  # noqa: E402
from conftest import load_golden  # noqa: E402
from gpu_util import DEV, bf, check, tiny_geo  # noqa: E402
from grounded_video_llm_amd import engine as E, weights as Wt, synth  # noqa: E402


def _clip_engine(c, seed):
    geo = tiny_geo(clip_hidden=c["hidden"], clip_inter=c["inter"], clip_layers=c["layers"], clip_heads=c["heads"], clip_image=c["image"],
                   clip_patch=c["patch"], max_segs=2)
    W = synth.clip_weights(c["hidden"], c["inter"], c["layers"], c["image"], c["patch"], seed=seed)
    eng = E.Engine(geo, DEV, towers=("clip",))
    eng.load_packed(Wt.pack_clip(W, c["layers"] - 1))
    eng.finalize()
    return eng, W


@pytest.mark.parametrize("name,tol_g,tol_o", [("clip_tiny", 8.8e-3, 7.4e-3), ("clip_full_layer", 5.8e-3, 3.9e-3)])   # observed 5.9e-3 / 5.0e-3, 3.8e-3 / 2.6e-3
def test_clip(name, tol_g, tol_o):
    meta, g = load_golden(name)
    c = meta["cfg"]
    eng, W = _clip_engine(c, meta["seed"])
    px = synth.det_tensor(meta["px"], meta["px_shape"])
    got = eng.clip_encode(px.to(DEV))
    st = meta.get("stride", [1, 1])
    check(got[:, ::st[0], ::st[1]], g["penultimate"], tol_g, f"{name} vs reference golden (fp32)")
    ref = O.clip_penultimate(px, W, c["layers"], c["heads"], emu=True)
    check(got, ref, tol_o, f"{name} vs oracle (bf16 emulation)")
    eng.debug_set("vision_in_place", 0)         # the round-2 operand path (Q / K pages + V^T transpose pass): bit-identical
    assert torch.equal(eng.clip_encode(px.to(DEV)), got)
    eng.debug_set("vision_in_place", 1)
    # round 4: width 1024 runs the patch embedding as ONE kernel (gvl_patch.hip); the three-pass path (patchify, GEMM, embed + LayerNorm) sums the same
    # products in another order: equal to fp32 rounding in front of the bf16 round of the conv output -- and both hold the golden bound above
    eng.debug_set("patch_fused", 0)
    three = eng.clip_encode(px.to(DEV))
    if c["hidden"] == 1024:
        nd = int((three != got).sum())
        print(f"[parity] {name}: fused patch embedding vs the three-pass path: {nd} of {got.numel()} values differ")
        check(got, three.float(), 6e-3, f"{name}: fused patch embedding vs the three-pass path")
        check(three[:, ::st[0], ::st[1]], g["penultimate"], tol_g, f"{name} (three-pass patch embedding) vs reference golden (fp32)")
    else:
        assert torch.equal(three, got)          # other widths: the knob changes nothing
    eng.close()


def _iv2_engine(c, seed, max_segs=2):
    geo = tiny_geo(iv2_dim=c["dim"], iv2_inter=c["inter"], iv2_depth=c["depth"], iv2_heads=c["heads"], iv2_image=c["image"],
                   frames_per_seg=c["frames"], max_segs=max_segs)
    W = synth.iv2_weights(c["dim"], c["inter"], c["depth"], c["frames"], c["image"], 14, seed=seed)
    eng = E.Engine(geo, DEV, towers=("iv2",))
    eng.load_packed(Wt.pack_iv2(W, c["depth"] - 1, c["frames"]))
    eng.finalize()
    return eng, W


@pytest.mark.parametrize("name,tol_g,tol_o", [("iv2_tiny", 1.8e-2, 9.9e-3), ("iv2_full_block", 1.26e-2, 6.1e-3)])   # observed 1.22e-2 / 6.6e-3, 8.4e-3 / 4.1e-3
def test_iv2(name, tol_g, tol_o):
    meta, g = load_golden(name)
    c = meta["cfg"]
    eng, W = _iv2_engine(c, meta["seed"])
    px = synth.det_tensor(meta["px"], meta["px_shape"])
    got = eng.iv2_encode(px.to(DEV))
    st = meta.get("stride", [1, 1])
    check(got[:, ::st[0], ::st[1]], g["out"], tol_g, f"{name} vs reference golden (fp32)")
    ref = O.iv2_encode(px, W, c["depth"], c["heads"], emu=True)
    check(got, ref, tol_o, f"{name} vs oracle (bf16 emulation)")
    # operand paths: 0 = Q / K pages + V^T transpose pass (round 2); 2 = V read in place -- the same arithmetic, bit-identical to 0;
    # 1 (default, `got`) = q in place too, normalised, scaled and rounded ONCE in the attention prologue with the softmax shift folded into
    # the S^T MFMAs: a different (not larger) set of rounding points, held to the same goldens above and to path 0 within bf16 noise
    eng.debug_set("attn_pipe", 0)              # the plain tile loop (attn_fwd_kernel) instead of the pipelined one: bit-identical
    assert torch.equal(eng.iv2_encode(px.to(DEV)), got)
    eng.debug_set("attn_pipe", 1)
    # round 5: `got` ran with the RMSNorms fused into the GEMMs around them (row statistics from the producing epilogue, norm weight folded into the
    # consuming weight, row scale in its epilogue): two activation roundings of the reference are gone, gamma * W is rounded once per weight.  The
    # separate norm pass (norm_fused = 0) is the arithmetic the oracle emulates; both are held to the reference golden and to each other.
    eng.debug_set("norm_fused", 0)
    unf = eng.iv2_encode(px.to(DEV))
    eng.debug_set("norm_fused", 1)
    print(f"[parity] {name}: fused vs unfused RMSNorm: {int((unf != got).sum())} of {got.numel()} values differ")
    check(unf[:, ::st[0], ::st[1]], g["out"], tol_g, f"{name} (separate norm pass) vs reference golden (fp32)")
    check(unf, ref, tol_o, f"{name} (separate norm pass) vs oracle (bf16 emulation)")
    check(got, unf.float(), tol_o, f"{name}: fused RMSNorm vs the separate norm pass")
    eng.debug_set("patch_fused", 0)            # round 4: width 1408 runs the patch embedding as ONE kernel; the three-pass path sums in another order
    three = eng.iv2_encode(px.to(DEV))
    eng.debug_set("patch_fused", 1)
    if c["dim"] == 1408:
        print(f"[parity] {name}: fused patch embedding vs the three-pass path: {int((three != got).sum())} of {got.numel()} values differ")
        check(got, three.float(), 8e-3, f"{name}: fused patch embedding vs the three-pass path")
        check(three[:, ::st[0], ::st[1]], g["out"], tol_g, f"{name} (three-pass patch embedding) vs reference golden (fp32)")
    else:
        assert torch.equal(three, got)
    eng.debug_set("vision_in_place", 0)
    paged = eng.iv2_encode(px.to(DEV))
    eng.debug_set("vision_in_place", 2)
    assert torch.equal(eng.iv2_encode(px.to(DEV)), paged)
    check(got, paged.float(), tol_o, f"{name}: folded in-place attention vs the paged path")
    check(paged[:, ::st[0], ::st[1]], g["out"], tol_g, f"{name} (paged path) vs reference golden (fp32)")
    eng.close()


@pytest.mark.parametrize("tower,hidden,heads", [("clip", 128, 2), ("clip", 192, 2), ("clip", 256, 2), ("iv2", 704, 8), ("iv2", 192, 2), ("iv2", 256, 2)])
def test_towers_at_other_head_dims(tower, hidden, heads):
    """The attention kernel's operand paths by head dim, at toy depth against the oracle (bf16 emulation): 64 (CLIP reads q, k, v in place),
    88 (InternVideo2: q in place with its RMSNorm in the prologue, softmax scale / shift folded into the S^T MFMAs, ones-column row sum),
    96 (V in place, no padding) and 128 (Q / K pages + V^T transpose pass -- the in-place V gather is built for 64 and 96 only); each also
    against the paged path (gvl_debug_set vision_in_place = 0)."""
    px_seed = f"hd.{tower}.{hidden}.{heads}"
    if tower == "clip":
        c = dict(hidden=hidden, inter=2 * hidden, layers=3, heads=heads, image=42, patch=14)          # S = 1 + 9
        eng, W = _clip_engine(c, px_seed)
        px = synth.det_tensor(px_seed + ".px", (2, 3, 42, 42))
        run = lambda: eng.clip_encode(px.to(DEV))
        ref = O.clip_penultimate(px, W, c["layers"], c["heads"], emu=True)
    else:
        c = dict(dim=hidden, inter=2 * hidden, depth=3, heads=heads, image=56, frames=2)               # S = 1 + 2 * 16
        eng, W = _iv2_engine(c, px_seed)
        px = synth.det_tensor(px_seed + ".px", (2, 3, 2, 56, 56))
        run = lambda: eng.iv2_encode(px.to(DEV))
        ref = O.iv2_encode(px, W, c["depth"], c["heads"], emu=True)
    got = run()
    check(got, ref, 1.2e-2, f"{tower} hidden {hidden} / {heads} heads (head dim {hidden // heads}) vs oracle (bf16 emulation)")
    eng.debug_set("vision_in_place", 0)
    paged = run()
    if tower == "iv2" and hidden // heads == 88:
        check(got, paged.float(), 1.2e-2, f"{tower} head dim 88: folded in-place attention vs the paged path")
    else:
        assert torch.equal(got, paged)
    eng.close()


@pytest.mark.parametrize("image,frames,sharp", [(56, 2, 1), (56, 4, 1), (56, 8, 3), (56, 12, 1), (56, 16, 12), (14, 63, 1), (14, 127, 3), (14, 191, 1), (42, 71, 12), (112, 9, 3), (112, 9, 1), (56, 16, 60), (112, 16, 1), (14, 255, 3), (14, 511, 12)])
def test_iv2_pipelined_attention_against_the_plain_kernel(image, frames, sharp):
    """Round 4: InternVideo2's attention runs a software-pipelined key-tile loop (attn_iv2_pipe_kernel: S^T of tile t+1 and P.V of tile t
    carry the exp2 + pack work of tile t in their shadow, every MFMA heading a fenced group of <= 3 VALU fillers).  Its normal pass keeps
    the FIRST tile's row maximum as the softmax reference for the whole row (no per-tile row max); attn_fwd_kernel (gvl_debug_set
    attn_pipe = 0) moves its reference only when a later tile exceeds it by 2^8, which ordinary data never does -- there the two issue the
    same MFMAs on the same values in the same order and must agree BIT FOR BIT, over every shape of the tile loop: S = frames *
    (image / 14)^2 + 1 gives 1 tile (partial), 2, 3, 4, 5 tiles with a one-key tail; image 14 gives whole tiles only (S = 64, 128, 192:
    no tail mask, the last V tile in the ring instead of its own slot); 10 whole tiles; 577 keys.
    sharp = 3 scales q_norm / k_norm so that later tiles DO exceed the first tile's maximum by more than 2^8: the plain kernel moves its
    reference, the pipelined one does not -- the same softmax up to the rounding of P to bf16 (a different, not larger, set of rounding points; on such peaked rows one flipped P ulp is
    visible: checked to the bound both hold against the oracle, 1.5e-2 of the output scale).  sharp = 12 (and 60) makes the scores so peaked that exp2(score - first-tile max) overflows: the row-sum check at the end of the pass
    sends the block through the SAFE pass (per-tile max + lazy reference = the plain kernel's arithmetic): finite, and equal to the plain
    kernel wherever it ran."""
    c = dict(dim=704, inter=1408, depth=3, heads=8, image=image, frames=frames)
    seed = f"pipe.{image}.{frames}"
    geo = tiny_geo(iv2_dim=c["dim"], iv2_inter=c["inter"], iv2_depth=c["depth"], iv2_heads=c["heads"], iv2_image=image, frames_per_seg=frames, max_segs=3)
    W = synth.iv2_weights(c["dim"], c["inter"], c["depth"], frames, image, 14, seed=seed)
    if sharp > 1:
        for i in range(c["depth"]):
            W[f"blocks.{i}.attn.q_norm.weight"] = W[f"blocks.{i}.attn.q_norm.weight"] * sharp
            W[f"blocks.{i}.attn.k_norm.weight"] = W[f"blocks.{i}.attn.k_norm.weight"] * sharp
    eng = E.Engine(geo, DEV, towers=("iv2",))
    eng.load_packed(Wt.pack_iv2(W, c["depth"] - 1, frames))
    eng.finalize()
    px = synth.det_tensor(seed + ".px", (3, 3, frames, image, image))
    S = frames * (image // 14) ** 2 + 1
    got = eng.iv2_encode(px.to(DEV))
    eng.debug_set("attn_pipe", 0)
    plain = eng.iv2_encode(px.to(DEV))
    eng.debug_set("attn_pipe", 2)                    # the pipelined kernel's SAFE pass alone (per-tile row max + lazy reference): the plain kernel's arithmetic
    safe = eng.iv2_encode(px.to(DEV))
    eng.debug_set("attn_pipe", 1)
    nsafe = int((safe != plain).sum())
    print(f"[parity] iv2 pipelined attention S = {S}, sharp x{sharp}: safe pass differs from the plain kernel in {nsafe} of {safe.numel()} values")
    assert nsafe == 0
    nf_got, nf_plain = int((~torch.isfinite(got.float())).sum()), int((~torch.isfinite(plain.float())).sum())
    assert nf_got == 0 and nf_plain == 0, f"S = {S}, sharp x{sharp}: non-finite outputs: pipelined {nf_got}, plain {nf_plain} of {got.numel()}"
    again = eng.iv2_encode(px.to(DEV))
    assert torch.equal(got, again), f"S = {S}, sharp x{sharp}: the pipelined kernel is not reproducible ({int((got != again).sum())} values differ between two runs)"
    # opt-in: whole 256-row query blocks on the 8-wave form of the kernel (half the DMA pieces per MFMA; measured slower), the rest on the 4-wave form;
    # a row's arithmetic does not depend on the form
    eng.debug_set("attn_pipe_rows", 256)
    eight = eng.iv2_encode(px.to(DEV))
    eng.debug_set("attn_pipe_rows", 128)
    assert torch.equal(got, eight), f"S = {S}, sharp x{sharp}: 8-wave and 4-wave forms differ in {int((got != eight).sum())} values"
    nbad = int((got != plain).sum())
    dmax = float((got.float() - plain.float()).abs().max() / plain.float().abs().max())
    print(f"[parity] iv2 pipelined attention S = {S} ({(S + 63) // 64} key tiles), sharp x{sharp}: {nbad} of {got.numel()} values differ from the plain kernel (max {dmax:.2e} of the scale)")
    if sharp == 1:
        assert nbad == 0, f"S = {S} ({(S + 63) // 64} key tiles): pipelined attention differs from the plain kernel in {nbad} of {got.numel()} values"
    else:
        check(got, plain.float(), 1.5e-2, f"iv2 pipelined attention, S = {S}, sharp x{sharp}, vs the plain kernel")
    if frames <= 16 and sharp <= 3:                  # against the oracle too (bf16 emulation); the long-sequence cases would take the CPU minutes
        ref = O.iv2_encode(px, W, c["depth"], c["heads"], emu=True)
        check(got, ref, 1.5e-2 if sharp > 1 else 1.2e-2, f"iv2 pipelined attention, S = {S}, sharp x{sharp}, vs oracle (bf16 emulation)")
    eng.close()


@pytest.mark.parametrize("name,tol_g,tol_o", [("glue_phi3_5", 9.2e-3, 6.0e-3), ("glue_llama3", 6.9e-3, 8.2e-3)])   # observed 6.1e-3 / 4.0e-3, 4.6e-3 / 5.4e-3
def test_encode_segments_and_splice(name, tol_g, tol_o):
    """encode_images + prepare_multimodal_inputs on the 2-segment skeleton."""
    meta, g = load_golden(name)
    llm, hid = meta["llm"], meta["hidden"]
    cc, vc = meta["clip"], meta["iv2"]
    geo = E.TowerGeometry(llm=llm, clip_hidden=cc["hidden"], clip_inter=cc["inter"], clip_layers=cc["layers"], clip_heads=cc["heads"],
                          iv2_dim=vc["dim"], iv2_inter=vc["inter"], iv2_depth=vc["depth"], iv2_heads=vc["heads"], frames_per_seg=vc["frames"],
                          hidden=hid, inter=128, layers=1, heads=hid // 128, kv_heads=hid // 128, vocab=50, max_seq=1024, max_segs=2, kv_pages=16,
                          max_prefill=1024, lm_head_bias=False, rope_orig_max_pos=0)
    Wc = synth.clip_weights(cc["hidden"], cc["inter"], cc["layers"], 336, 14, seed="g.glue.clip")
    Wv = synth.iv2_weights(vc["dim"], vc["inter"], vc["depth"], vc["frames"], 224, 14, seed="g.glue.iv2")
    Wp = synth.projector_weights(llm, hid, 1024, 1408, seed="g.glue.proj." + llm)
    emb_w = synth.det_tensor("g.glue.embed." + llm, (50, hid), 0.5)
    Wl = synth.llm_weights(geo.kind, hid, 128, 1, hid // 128, hid // 128, 50, False, seed="g.glue.llm")
    Wl["model.embed_tokens.weight"] = emb_w
    eng = E.Engine(geo, DEV, towers=("clip", "iv2", "llm"))
    eng.load_packed(Wt.pack_clip(Wc, cc["layers"] - 1))
    eng.load_packed(Wt.pack_iv2(Wv, vc["depth"] - 1, vc["frames"]))
    eng.load_packed(Wt.pack_projectors(Wp, llm))
    eng.load_packed(Wt.pack_llm(Wl, geo.kind, 1, geo.heads, geo.kv_heads, geo.max_seq, geo.rope_theta))
    eng.finalize()
    assert eng.tokens_per_seg == (156 if llm == "phi3.5" else 64) + 16 * vc["frames"] + 1      # 285 / 193 at 8 frames per segment
    sp = synth.det_tensor("g.glue.sp", (1, 2, 3, 336, 336))
    tp = synth.det_tensor("g.glue.tp", (1, 4, 3, 224, 224))
    tseg = tp.reshape(1, 2, 2, 3, 224, 224).permute(0, 1, 3, 2, 4, 5).flatten(0, 1)        # (b s) c f h w
    vis = eng.encode_segments(sp[0].to(DEV), tseg.to(DEV))
    assert list(vis.shape) == meta["feats_shape"][1:]
    s = meta["stride"]
    check(vis[None][:, ::s[0], ::s[1]], g["feats"], tol_g, f"{name}: encode_images vs reference golden")
    ref = O.encode_images(sp, tp, Wc, Wv, Wp, llm, clip_layers=cc["layers"], clip_heads=cc["heads"], iv2_depth=vc["depth"], iv2_heads=vc["heads"], emu=True)
    check(vis, ref[0], tol_o, f"{name}: encode_images vs oracle (bf16 emulation)")
    emb = eng.splice(meta["ids"], vis)
    assert list(emb.shape) == meta["emb_shape"][1:]
    check(emb[None][:, ::s[0], ::s[1]], g["emb"], tol_g, f"{name}: spliced inputs_embeds vs reference golden")
    # the text rows are pure gathers: bit-exact against the bf16 embedding table
    idx = meta["ids"].index(-200)
    assert torch.equal(emb[:idx].cpu(), emb_w.to(bf)[meta["ids"][:idx]])
    assert torch.equal(emb[idx + vis.shape[0]:].cpu(), emb_w.to(bf)[meta["ids"][idx + 1:]])
    assert torch.equal(emb[idx: idx + vis.shape[0]], vis)
    eng.close()
