"""Frame preprocessing: eva_clip.py:125-153 = Resize(BICUBIC) -> CenterCrop -> ToTensor ->
Normalize.  CPU part: the numpy oracle is pinned bit-exactly to Pillow itself and to the committed digests
(tests/golden/preprocess.json, generated with the real Pillow by tests/golden/make_golden.py), and the C-ABI's
host-side plan (bounds + fixed-point weights) equals the oracle's tables.  GPU part: the HIP kernels reproduce the
oracle byte for byte through the C ABI."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

from hirest_amd import synth
from oracle import preprocess_cpu as P

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "preprocess.json")))
CASES = sorted(GOLD["cases"])


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _frame(name):
    c = GOLD["cases"][name]
    return synth.rgb_frames("preprocess." + name, (c["H"], c["W"], 3), 5), c


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_golden_digests(name):
    arr, c = _frame(name)
    nw, nh = P.resized_size(c["W"], c["H"], c["size"])
    assert [nw, nh] == c["resized"] and list(P.crop_origin(nw, nh, c["size"])) == c["crop"]
    u8 = P.transform_u8(arr, c["size"])
    assert u8[0, :8].tolist() == c["row0_u8"]
    assert _sha(u8) == c["sha256_u8"]
    assert _sha(P.to_tensor_normalize(u8)) == c["sha256_f32"]


@pytest.mark.parametrize("hw", [(97, 131), (300, 200), (64, 64), (500, 1000), (1000, 225)])
def test_oracle_matches_pillow_live(hw):
    Image = pytest.importorskip("PIL.Image")
    h, w = hw
    arr = synth.rgb_frames("preprocess.live", (h, w, 3), h * 1000 + w)
    for (ow, oh) in [(224, 224), P.resized_size(w, h, 224), (w // 2 + 1, h * 2), (w, h // 3 + 1)]:
        ref = np.asarray(Image.fromarray(arr).resize((ow, oh), Image.BICUBIC))
        assert np.array_equal(P.resize_bicubic_u8(arr, ow, oh), ref), (hw, ow, oh)


def test_cpu_image_transform_equals_oracle():
    """hirest_amd.image_transform (the per-image PIL path kept for PIL inputs) == the oracle's whole transform."""
    Image = pytest.importorskip("PIL.Image")
    import hirest_amd
    arr, c = _frame("odd")
    t = hirest_amd.image_transform(224)(Image.fromarray(arr))
    assert t.dtype == torch.float32 and tuple(t.shape) == (3, 224, 224)
    assert np.array_equal(t.numpy(), P.image_transform(arr, 224))


def _split_plan(pl, S):
    ksh, ksv = int(pl[7]), int(pl[8])
    hb = pl[16:16 + 2 * S].reshape(S, 2)
    hk = pl[16 + 2 * S:16 + 2 * S + ksh * S].reshape(ksh, S).T
    o = 16 + 2 * S + ksh * S
    vb = pl[o:o + 2 * S].reshape(S, 2)
    vk = pl[o + 2 * S:o + 2 * S + S * ksv].reshape(S, ksv)
    return hb, hk, vb, vk


@pytest.mark.parametrize("hws", [(360, 640, 224), (640, 360, 224), (224, 224, 224), (1080, 1920, 224), (333, 500, 224),
                                 (120, 160, 224), (250, 224, 224), (37, 53, 28), (2160, 3840, 224)])
def test_host_plan_equals_oracle_tables(hws):
    """hirest_preprocess_plan runs on the host (no GPU needed): geometry and every fixed-point weight must equal the
    oracle's restatement of Pillow's precompute_coeffs / normalize_coeffs_8bpc."""
    from hirest_amd.preprocess import host_plan
    h, w, S = hws
    pl = host_plan(h, w, S)
    nw, nh = P.resized_size(w, h, S)
    left, top = P.crop_origin(nw, nh, S)
    assert tuple(int(v) for v in pl[:7]) == (h, w, S, nw, nh, left, top)
    hb, hk, vb, vk = _split_plan(pl, S)
    for ins, outs, first, b, k in ((w, nw, left, hb, hk), (h, nh, top, vb, vk)):
        if ins == outs:
            assert k.shape[1] == 1 and (b[:, 0] == np.arange(first, first + S)).all() and (k == 1 << 22).all()
            continue
        rb, rk = P.resample_coeffs(ins, outs)
        assert rk.shape[1] == k.shape[1]
        assert np.array_equal(rb[first:first + S], b) and np.array_equal(rk[first:first + S], k)
    # rows / columns the device passes touch
    assert int(pl[9]) == vb[0, 0] and int(pl[9] + pl[10]) == vb[-1, 0] + vb[-1, 1]
    assert int(pl[11]) == hb[0, 0] and int(pl[11] + pl[12]) == hb[-1, 0] + hb[-1, 1]


def test_plan_argument_errors():
    from hirest_amd import _lib
    lib = _lib.load()
    assert lib.hirest_preprocess_plan_bytes(0, 10, 224) < 0
    assert lib.hirest_preprocess_plan(360, 640, 224, None, 0) == -1                    # HIREST_E_BADARG
    buf = np.zeros(4, dtype=np.int32)
    assert lib.hirest_preprocess_plan(360, 640, 224, buf.ctypes.data, 16) == -3        # HIREST_E_WORKSPACE: blob too small


def test_no_cpu_fallback():
    from hirest_amd.preprocess import FramePreprocessor
    with pytest.raises(RuntimeError):
        FramePreprocessor(224)(torch.zeros((1, 32, 32, 3), dtype=torch.uint8))
    with pytest.raises(ValueError):
        FramePreprocessor(224)(torch.zeros((1, 3, 32, 32), dtype=torch.float32))


# ------------------------------------------------------------------------------------------------ GPU

@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_gpu_preprocess_bit_exact(name):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from hirest_amd.preprocess import FramePreprocessor
    arr, c = _frame(name)
    S = c["size"]
    B = 3
    batch = np.stack([arr, arr[::-1].copy(), synth.rgb_frames("preprocess.b2." + name, arr.shape, 9)])
    pre = FramePreprocessor(S)
    dev = torch.device("cuda:0")
    x = torch.from_numpy(batch).to(dev)
    u8 = pre(x).cpu().numpy()
    f32 = pre(x, normalized=True).cpu().numpy()
    assert u8.shape == (B, S, S, 3) and f32.shape == (B, 3, S, S)
    assert _sha(u8[0]) == c["sha256_u8"]                      # == the real Pillow pipeline
    for b in range(B):
        ref = P.transform_u8(batch[b], S)
        assert np.array_equal(u8[b], ref), f"frame {b}: {np.abs(u8[b].astype(int) - ref).max()}"
        assert np.array_equal(f32[b], P.to_tensor_normalize(ref))
    assert _sha(f32[0]) == c["sha256_f32"]


@pytest.mark.gpu
def test_gpu_preprocess_unaligned_views_and_properties():
    """Row base addresses with every alignment (odd widths, sliced batches); constant frames stay constant;
    a 224x224 frame passes through untouched; flipping the input flips the output (filter symmetry)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from hirest_amd.preprocess import FramePreprocessor
    dev = torch.device("cuda:0")
    pre = FramePreprocessor(224)
    for (h, w) in [(241, 323), (242, 325), (243, 330)]:
        arr = synth.rgb_frames("preprocess.align", (5, h, w, 3), h)
        x = torch.from_numpy(arr).to(dev)
        out = pre(x[1:4]).cpu().numpy()                        # sliced view: base pointer offset by one odd-sized frame
        for i in range(3):
            assert np.array_equal(out[i], P.transform_u8(arr[1 + i], 224))
    pre30 = FramePreprocessor(30)                              # 90-byte rows: the byte-wise vertical kernel
    arr = synth.rgb_frames("preprocess.s30", (2, 77, 131, 3), 4)
    out = pre30(torch.from_numpy(arr).to(dev)).cpu().numpy()
    f32 = pre30(torch.from_numpy(arr).to(dev), normalized=True).cpu().numpy()
    for i in range(2):
        assert np.array_equal(out[i], P.transform_u8(arr[i], 30))
        assert np.array_equal(f32[i], P.to_tensor_normalize(out[i]))
    const = torch.full((2, 360, 640, 3), 201, dtype=torch.uint8, device=dev)
    assert (pre(const) == 201).all()
    same = torch.from_numpy(synth.rgb_frames("preprocess.same", (2, 224, 224, 3), 1)).to(dev)
    assert torch.equal(pre(same), same)
    arr = torch.from_numpy(synth.rgb_frames("preprocess.flip", (1, 360, 640, 3), 2)).to(dev)
    a = pre(arr)
    b = pre(torch.flip(arr, dims=(1, 2)))
    assert torch.equal(torch.flip(b, dims=(1, 2)), a)


@pytest.mark.gpu
def test_gpu_preprocess_full_size_batch_and_encode():
    """1080p batch (BASELINE-size input geometry): exact vs the oracle on sampled frames, and the uint8 crop feeds
    encode_image (fused normalisation) with the same embeddings as the reference-style fp32 tensor."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import hirest_amd
    from hirest_amd.preprocess import FramePreprocessor
    dev = torch.device("cuda:0")
    B = 48
    arr = synth.rgb_frames("preprocess.big", (B, 1080, 1920, 3), 3)
    pre = FramePreprocessor(224)
    x = torch.from_numpy(arr).to(dev)
    u8 = pre(x)
    for b in (0, 17, 47):
        assert np.array_equal(u8[b].cpu().numpy(), P.transform_u8(arr[b], 224))
    model, _ = hirest_amd.build_eva_model_and_transforms("EVA_CLIP_tiny_test", pretrained="synth:11", precision="bf16")
    model = model.to(dev).eval()
    e_u8 = model.encode_image(u8[:8])
    e_f32 = model.encode_image(pre(x[:8], normalized=True))
    cos = torch.nn.functional.cosine_similarity(e_u8, e_f32, dim=-1)
    assert cos.min().item() > 0.9999
