"""-m gpu: frame pre-processing on the GPU through the C ABI -- BIT-EXACT (it is integer / byte work plus three IEEE
f32 operations) against the images Pillow produced (tests/golden/preprocess.npz) and against the oracle on seeded ragged sizes."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import gvl_oracle as O  # noqa: E402
from conftest import load_golden  # noqa: E402
from gpu_util import DEV  # noqa: E402
from grounded_video_llm_amd import engine as E, lib as L  # noqa: E402


@pytest.fixture(scope="module")
def eng():
    e = E.Engine(E.TowerGeometry(max_segs=1), DEV, towers=())
    yield e
    e.close()


def _expect(u8_hwc, mean, std):
    x = np.transpose(u8_hwc, (2, 0, 1)).astype(np.float32) / np.float32(255)
    return (x - np.asarray(mean, np.float32).reshape(3, 1, 1)) / np.asarray(std, np.float32).reshape(3, 1, 1)


@pytest.mark.parametrize("layout", ["hwc", "chw"])
def test_preprocess_matches_pillow_golden(eng, layout):
    meta, g = load_golden("preprocess")
    for name, c in meta["cases"].items():
        img = O.synthetic_frame(name, c["h"], c["w"])
        fr = torch.from_numpy(img)[None]
        if layout == "chw":
            fr = fr.permute(0, 3, 1, 2).contiguous()
        for mean, std in ((O.INTERNVIDEO_MEAN, O.INTERNVIDEO_STD), (O.OPENAI_DATASET_MEAN, O.OPENAI_DATASET_STD)):
            got = eng.preprocess_frames(fr.to(DEV), c["size"], mean, std).cpu().numpy()[0]
            ref = _expect(g[name], mean, std)
            assert got.shape == ref.shape and np.array_equal(got, ref), f"{name}/{layout}: {int((got != ref).sum())} of {ref.size} values differ from Pillow"


def test_preprocess_ragged_batch_vs_oracle(eng):
    """several frames per call, sizes that exercise both crop roundings, tiny and 1-pixel-margin cases, the up-scaling branch, identity."""
    rs = np.random.RandomState(5)
    for h, w, size, n in ((229, 224, 224, 3), (224, 229, 224, 2), (37, 640, 32, 2), (640, 37, 32, 2), (336, 336, 336, 2), (100, 161, 224, 2),
                          (481, 853, 336, 4), (33, 33, 32, 1), (225, 224, 224, 1)):
        frames = rs.randint(0, 256, (n, h, w, 3)).astype(np.uint8)
        got = eng.preprocess_frames(torch.from_numpy(frames).to(DEV), size, O.OPENAI_DATASET_MEAN, O.OPENAI_DATASET_STD).cpu().numpy()
        for i in range(n):
            ref = O.frame_transform(np.transpose(frames[i], (2, 0, 1)), size, O.OPENAI_DATASET_MEAN, O.OPENAI_DATASET_STD)
            assert np.array_equal(got[i], ref), f"{h}x{w}->{size} frame {i}: {int((got[i] != ref).sum())} values differ"


def test_preprocess_clip_shapes_properties(eng):
    """full BASELINE size (96 frames of 360p -> 224, 12 frames -> 336): constant frames stay constant (the taps sum to 2^22 exactly
    enough for clip8 to return the input), and the batch is frame-independent (frame i alone == frame i in the batch)."""
    n, h, w = 96, 360, 640
    g = torch.Generator(device="cpu"); g.manual_seed(11)
    frames = torch.randint(0, 256, (n, h, w, 3), generator=g, dtype=torch.uint8)
    frames[0] = 200
    frames[1] = 0
    frames[2] = 255
    out = eng.preprocess_frames(frames.to(DEV), 224, O.INTERNVIDEO_MEAN, O.INTERNVIDEO_STD)
    assert out.shape == (n, 3, 224, 224)
    for i, v in ((0, 200), (1, 0), (2, 255)):
        ref = _expect(np.full((224, 224, 3), v, np.uint8), O.INTERNVIDEO_MEAN, O.INTERNVIDEO_STD)
        assert np.array_equal(out[i].cpu().numpy(), ref)
    one = eng.preprocess_frames(frames[40:41].to(DEV), 224, O.INTERNVIDEO_MEAN, O.INTERNVIDEO_STD)
    assert torch.equal(one[0], out[40])
    sp = eng.preprocess_frames(frames[4::8].to(DEV), 336, O.OPENAI_DATASET_MEAN, O.OPENAI_DATASET_STD)
    assert sp.shape == (12, 3, 336, 336)
    ref = O.frame_transform(frames[4].permute(2, 0, 1).numpy(), 336, O.OPENAI_DATASET_MEAN, O.OPENAI_DATASET_STD)
    assert np.array_equal(sp[0].cpu().numpy(), ref)


def test_preprocess_bad_arguments(eng):
    fr = torch.zeros((1, 8, 8, 3), dtype=torch.uint8, device=DEV)
    with pytest.raises(L.GvlError):
        eng.preprocess_frames(fr, 0, (0, 0, 0), (1, 1, 1))
    with pytest.raises(L.GvlError):
        eng.preprocess_frames(fr, 8, (0, 0, 0), (1, 0, 1))


def test_preprocess_random_geometries_vs_oracle(eng):
    """Random source sizes / target sizes / layouts (strong down-scaling, up-scaling, thin frames, saturated edges): the GPU path
    against the oracle restatement (itself checked live against Pillow on CPU, tests/test_oracle_golden.py) -- bit-exact."""
    rng = np.random.default_rng(11)
    geoms = [(1080, 1920, 224), (17, 640, 16), (37, 41, 64), (224, 224, 224), (5, 300, 4), (400, 226, 224), (2, 2, 8)]
    geoms += [(int(rng.integers(2, 500)), int(rng.integers(2, 500)), int(rng.integers(1, 240))) for _ in range(14)]
    for h, w, size in geoms:
        n = int(rng.integers(1, 4))
        img = rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)
        img[:, : h // 2, : w // 2] = 255
        img[:, h // 2:, w // 2:] = 0
        for layout in ("hwc", "chw"):
            fr = torch.from_numpy(img)
            if layout == "chw":
                fr = fr.permute(0, 3, 1, 2).contiguous()
            got = eng.preprocess_frames(fr.to(DEV), size, O.OPENAI_DATASET_MEAN, O.OPENAI_DATASET_STD).cpu().numpy()
            for i in range(n):
                ref = O.frame_transform(np.transpose(img[i], (2, 0, 1)), size, O.OPENAI_DATASET_MEAN, O.OPENAI_DATASET_STD)
                assert got[i].shape == ref.shape and np.array_equal(got[i], ref), f"{(h, w)} -> {size} {layout}: {int((got[i] != ref).sum())} values differ"
