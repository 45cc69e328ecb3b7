"""CPU-side checks: the C-ABI library builds/loads and exports every symbol the header declares
(no compute calls without a GPU), and the host layer mirrors the reference interface."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from hirest_amd import _lib, build
    build.build(verbose=False)
    return _lib.load()


def test_header_symbols_all_exported(lib):
    from hirest_amd import _lib
    hdr = open(os.path.join(REPO, "include", "hirest_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(hirest_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 18
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), f"{name} declared in include/hirest_hip.h but not exported"
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    assert lib.hirest_abi_version() == 4 == _lib.ABI_VERSION
    assert b"gfx950" in lib.hirest_build_info()


def test_argument_errors_without_gpu(lib):
    from hirest_amd import _lib
    a = _lib.GemmArgs()  # all-NULL
    assert lib.hirest_gemm_bf16(ctypes.byref(a), None) == -1
    # a binding built against another struct layout (ABI 2 had no struct_size / flags) is rejected before any member
    # behind its end is read: non-NULL operands, valid shape, wrong struct_size -> BADARG, never a launch
    b = _lib.GemmArgs(ctypes.sizeof(_lib.GemmArgs) - 8, 1 << 20, 64, 1 << 21, 64, None, 1 << 22, 64, 64, 64, 64, 0)
    assert lib.hirest_gemm_bf16(ctypes.byref(b), None) == -1
    assert _lib.GemmArgs.make().struct_size == ctypes.sizeof(_lib.GemmArgs) == 120
    assert lib.hirest_layernorm(None, 0, None, None, None, 0.0, None, 0, 0, 0, 0, None) == -1
    assert lib.hirest_attention_bf16(None, None, 1, 1, 1, 64, 1.0, 0, None) == -1
    assert lib.hirest_vision_workspace_bytes(None, 4) == 0
    # round-2 entry points: captioning step, beam bookkeeping, ragged BERT batch, fp32 GEMM selector
    d = _lib.CaptionDecoder(2, 12, 768, 3072, 30528, 512)
    al = lambda v: (v + 255) // 256 * 256
    assert lib.hirest_caption_step_workspace_bytes(ctypes.byref(d), 25) == \
        5 * al(25 * 768 * 4) + al(25 * 3 * 768 * 4) - al(25 * 768 * 4) + al(25 * 3072 * 4) + al(25 * 30528 * 4) + al(25 * 4) + al(25 * 1908 * 4)
    assert lib.hirest_caption_step_workspace_bytes(None, 25) == 0
    assert lib.hirest_caption_decode_step(ctypes.byref(d), 25, 0, None, None, None, None, None, 20, None, None, None, 0, None) == -1
    assert lib.hirest_beam_advance(None, None, 5, 5, 30528, 0, 48, 102, None, None, None, None, None, None, None, None, None) == -1
    assert lib.hirest_attention_f32_varlen(None, None, None, 1, 1, 12, 64, 0.125, 0.0, None) == -1
    assert lib.hirest_pool_l2norm_varlen(None, None, None, 1, 384, None) == -1
    assert lib.hirest_embedding_pos_fwd_f32(None, None, None, None, None, 1, 384, None) == -1
    assert lib.hirest_gemm_f32_select_kernel(3) == -1 and lib.hirest_gemm_f32_select_kernel(0) == 0
    # round 3: the step-captioning decode kernels
    assert lib.hirest_caption_select(2) == -1 and lib.hirest_caption_select(0) == 0
    assert lib.hirest_caption_decode_logits(ctypes.byref(d), 25, 0, None, None, None, None, None, 20, None, None, 0, None) == -1
    assert lib.hirest_caption_beam_tail_workspace_bytes(5, 5, 30528) == 25 * 8 * 5 * 8 + 25 * 17 * 4     # candidates (score, index) + row statistics
    assert lib.hirest_caption_beam_tail_workspace_bytes(0, 5, 30528) == 0
    assert lib.hirest_caption_beam_tail(None, 30528, None, 5, 5, 30528, 0, 48, 102, None, None, None, None, None, None, None, None, None, None, 0,
                                        None) == -1
    assert lib.hirest_gemm_f32_ln(None, 768, None, None, None, None, None, 1e-12, None, 0, None, 768, None, None, 0, None, 768, 25, 768, 768, 0,
                                  None) == -1
    assert lib.hirest_attention_f32_decode(None, 768, None, None, 768, None, 0, None, None, 0, None, None, None, 25, 12, 0.125, 0.0, 0.0, None) == -1
    assert lib.hirest_caption_beam_step(ctypes.byref(d), 5, 5, 0, None, None, None, None, None, 20, None, None, 48, 102, None, None, None, None,
                                        None, None, None, 0, None, 0, None) == -1
    assert lib.hirest_gemm_f32_ln_colmax(None, 768, None, None, 1e-12, None, 768, None, None, 30528, None, 25, 30528, 768, None) == -1
    assert lib.hirest_gemm_f32_workspace_bytes(1500, 768, 3072) == 4 * 1500 * 768 * 4       # 288 tiles: split over the four K quarters
    assert lib.hirest_gemm_f32_workspace_bytes(1500, 3072, 768) == 0 and lib.hirest_gemm_f32_workspace_bytes(25, 768, 3072) == 0
    assert lib.hirest_gemm_f32_ws(None, 768, None, 768, None, None, 0, None, 0, None, 768, 1500, 768, 3072, 0, None, 0, None) == -1
    assert lib.hirest_gemm_f32_layouts(None, 768, 1, None, 768, 1, None, None, 0, None, 768, 768, 768, 1500, 0, None, 0, None) == -1
    assert lib.hirest_gemm_f32_layouts_workspace_bytes(768, 768, 1500) == 4 * 768 * 768 * 4 and lib.hirest_gemm_f32_layouts_workspace_bytes(3072, 3072, 1500) == 0
    assert lib.hirest_joint_time_grid_f32(None, 5, 300, None, None) == -1
    from hirest_amd._lib import ColsumItem, COLSUM_GROUP_MAX
    assert ctypes.sizeof(ColsumItem) == 56 and COLSUM_GROUP_MAX == 40       # hirest_colsum_item / HIREST_COLSUM_GROUP_MAX
    assert lib.hirest_weighted_colsum_grouped_f32(None, 3, None) == -1
    assert lib.hirest_weighted_colsum_grouped_f32((ColsumItem * 2)(), 2, None) == -1     # NULL matrices: refused before any launch
    # round 6: the joint model's bf16x3 pieces, the C-issued training step, the captioning read-out
    assert lib.hirest_layernorm_f32_split2(None, 768, None, 0, None, None, 1e-12, None, 768, None, 1536, 5, 768, None) == -1
    assert lib.hirest_split2_transposed_bf16(None, 768, None, 1536, 768, 768, None) == -1
    assert ctypes.sizeof(_lib.SplitItem) == 48
    assert lib.hirest_split2_grouped_bf16(None, 2, None) == -1 and lib.hirest_split2_grouped_bf16((_lib.SplitItem * 2)(), 2, None) == -1
    assert lib.hirest_beam_backtrack(None, None, None, None, 5, 5, 48, None, None) == -1
    tb = _lib.TrainBlock()
    assert lib.hirest_train_block_forward_scratch_bytes(ctypes.byref(tb)) == 0                 # struct_size 0: another layout
    tb.struct_size = ctypes.sizeof(_lib.TrainBlock)
    tb.B, tb.T, tb.heads, tb.width, tb.mlp = 5, 300, 12, 768, 3072
    al64 = lambda n: (n + 63) // 64 * 64
    R = 1500
    assert lib.hirest_train_block_forward_scratch_bytes(ctypes.byref(tb)) == 4 * al64(R * 768)
    assert lib.hirest_train_block_backward_scratch_bytes(ctypes.byref(tb)) == \
        4 * (8 * al64(R * 768) + 2 * al64(R * 3072) + al64(5 * 12 * 300 * 300) + al64(R * 2304))
    tb.precision = 1
    assert lib.hirest_train_block_forward_scratch_bytes(ctypes.byref(tb)) > 4 * al64(R * 768)
    tb.precision = 2
    assert lib.hirest_train_block_forward_scratch_bytes(ctypes.byref(tb)) == 0                 # unknown precision
    tb.precision = 0
    assert lib.hirest_train_block_forward(ctypes.byref(tb), None, 0, None) == -1               # no scratch: refused before any launch
    tg = _lib.TrainBlockGrads()
    assert lib.hirest_train_block_backward(ctypes.byref(tb), ctypes.byref(tg), None) == -1     # struct_size 0
    tf = _lib.TrainFusionBwd()
    assert lib.hirest_train_fusion_backward_scratch_bytes(ctypes.byref(tf)) == 0 and lib.hirest_train_fusion_backward(ctypes.byref(tf), None) == -1
    tf.struct_size = ctypes.sizeof(_lib.TrainFusionBwd)
    tf.B, tf.T, tf.E, tf.W, tf.vis_dim, tf.text_dim, tf.asr_dim, tf.max_pos = 5, 300, 512, 768, 1024, 1024, 384, 2048
    assert lib.hirest_train_fusion_backward_scratch_bytes(ctypes.byref(tf)) > 0
    assert lib.hirest_train_fusion_backward(ctypes.byref(tf), None) == -1                      # no dx / scratch
    tf.T = 4096
    assert lib.hirest_train_fusion_backward_scratch_bytes(ctypes.byref(tf)) == 0               # longer than the position table


def test_workspace_size_formula(lib):
    from hirest_amd import _lib
    t = _lib.VisionTower()
    t.image_size, t.patch, t.width, t.heads, t.head_dim, t.mlp_dim, t.layers, t.embed_dim, t.kpad = 224, 14, 1408, 16, 88, 6144, 40, 1024, 640
    M = 256 * 257
    al = lambda v: (v + 255) // 256 * 256
    assert lib.hirest_vision_workspace_bytes(ctypes.byref(t), 256) == al(M * 1408 * 4) + al(M * 1408 * 2) + al(M * 6144 * 2) + al(256 * 4)


def test_model_schema_and_errors():
    import hirest_amd
    from hirest_amd import synth
    with pytest.raises(RuntimeError):
        hirest_amd.build_eva_model_and_transforms("no-such-model", pretrained="synth:0")
    with pytest.raises(Exception):
        hirest_amd.build_eva_model_and_transforms("EVA_CLIP_tiny_test", pretrained="/nonexistent.pt")
    model, pre = hirest_amd.build_eva_model_and_transforms("EVA_CLIP_tiny_test", pretrained="synth:11", precision="bf16")
    assert model.training  # the reference returns a train-mode module (eva_clip.py:155-172)
    sd = synth.eva_clip_state_dict(synth.EVA_CLIP_TINY, 11)
    got = model.state_dict()
    assert set(got) == set(sd) and all(torch.equal(got[k], sd[k]) for k in sd)
    assert model.visual.image_size == 224 and model.visual.image_mean == hirest_amd.eva_clip.OPENAI_DATASET_MEAN
    for p in model.parameters():
        p.requires_grad = False           # modeling.py:126-129 freeze loop works
    assert sum(p.numel() for p in model.parameters()) == sum(v.numel() for v in sd.values())
    with pytest.raises(AssertionError):   # vit_model.py:203
        model.encode_image(torch.zeros(1, 3, 200, 224))
    with pytest.raises(RuntimeError):     # no silent CPU fallback
        model.encode_image(torch.zeros(1, 3, 224, 224))
    with pytest.raises(RuntimeError):
        model.encode_text(torch.zeros(1, 77, dtype=torch.long))
    # checkpoint round trip incl. the 'module.' prefix + wrapper keys (eva_clip.py:68-79)
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "ck.pt")
        torch.save({"model": {"module." + k: v for k, v in sd.items()}}, path)
        m2, _ = hirest_amd.build_eva_model_and_transforms("EVA_CLIP_tiny_test", pretrained=path)
        assert all(torch.equal(m2.state_dict()[k], sd[k]) for k in sd)


def test_eva_g14_param_count():
    """1 012 588 928 vision + 123 846 913 text params, without allocating them."""
    from hirest_amd import synth
    nv = sum(int(np.prod(s)) for s in synth.eva_vision_shapes(synth.EVA_CLIP_G_14).values())
    nt = sum(int(np.prod(s)) if len(s) else 1 for s in synth.eva_text_shapes(synth.EVA_CLIP_G_14).values())
    assert nv == 1012588928 and nt == 123846913


def test_image_transform_geometry():
    from PIL import Image
    import hirest_amd
    tf = hirest_amd.image_transform(224)
    rng = np.random.RandomState(0)
    for (h, w) in [(300, 400), (400, 300), (224, 224), (225, 1000)]:
        img = Image.fromarray(rng.randint(0, 255, (h, w, 3), dtype=np.uint8))
        t = tf(img)
        assert t.shape == (3, 224, 224) and t.dtype == torch.float32
    # a 224x224 image is only converted + normalised: exact formula check
    arr = rng.randint(0, 255, (224, 224, 3), dtype=np.uint8)
    t = tf(Image.fromarray(arr))
    mean = np.array(hirest_amd.eva_clip.OPENAI_DATASET_MEAN, dtype=np.float32).reshape(3, 1, 1)
    std = np.array(hirest_amd.eva_clip.OPENAI_DATASET_STD, dtype=np.float32).reshape(3, 1, 1)
    want = (arr.astype(np.float32).transpose(2, 0, 1) / 255.0 - mean) / std
    assert np.allclose(t.numpy(), want, atol=1e-6)


def test_moment_model_builds_its_own_clip_like_the_reference(tmp_path, monkeypatch):
    """modeling.py:115-129: MomentModel(n_frames, asr_dim, args) builds EVA_CLIP_g_14 from ./pretrained_weights/
    eva_clip_psz14.pt, casts to float, puts it in eval mode and freezes it.  Offline the two args fields redirect the build
    to a synthetic checkpoint of the tiny towers; without them the reference's file is looked for and its absence is the
    reference's error."""
    import hirest_amd
    from hirest_amd import synth

    class Args:
        clip_model_name = "EVA_CLIP_tiny_e1024_test"
        clip_pretrained = "synth:11"
    m = hirest_amd.MomentModel(n_frames=-1, asr_dim=384, args=Args())
    assert isinstance(m.clip_model, hirest_amd.EVA_CLIP) and not m.clip_model.training
    assert all(not p.requires_grad for p in m.clip_model.parameters())
    assert all(p.dtype == torch.float32 for p in m.clip_model.parameters())
    assert any(p.requires_grad for n, p in m.named_parameters() if not n.startswith("clip_model."))
    sd = m.state_dict()
    assert "clip_model.text.token_embedding.weight" in sd and "clip_model.visual.blocks.0.attn.q_bias" in sd
    want = synth.eva_clip_state_dict(dict(synth.EVA_CLIP_TINY, embed_dim=1024), 11)
    assert torch.equal(sd["clip_model.text.text_projection"], want["text.text_projection"])
    assert m.clip_preprocess is not None
    with pytest.raises(RuntimeError):                       # no CPU fallback: the text tower refuses host tensors
        m.test_step({"tasks": ["moment_retrieval"], "clip_text_ids": torch.zeros(1, 77, dtype=torch.long),
                     "vis_feats": torch.zeros(1, 4, 1024), "vis_mask": torch.ones(1, 4, dtype=torch.long),
                     "moment_mask": torch.ones(1, 4, dtype=torch.long), "asr_feats": torch.zeros(1, 4, 384)})
    monkeypatch.chdir(tmp_path)                             # no ./pretrained_weights here
    with pytest.raises(FileNotFoundError):
        hirest_amd.MomentModel(n_frames=-1, asr_dim=384, args=None)
    assert hirest_amd.MomentModel(n_frames=-1, asr_dim=384, args=None, clip_model=None).clip_model is None


def test_profiled_kernels_are_the_dispatched_ones(lib):
    """a committed profile must describe kernels the library launches TODAY.  Every gemm_* instantiation named in
    the newest profiles/rNN/pmc_traffic.json and pmc_summary.md (section 1: the bench's kernels) has to be what
    hirest_gemm_dispatch_name returns for one of the tower's problems under the default selection; and bench.py's roofline
    looks its traffic figure up in that same directory."""
    import json
    from hirest_amd import _lib

    def name(M, N, K, epi):
        a = _lib.GemmArgs.make(1, K, 1, K, None, 1, N, M, N, K, epi)
        buf = ctypes.create_string_buffer(64)
        assert lib.hirest_gemm_dispatch_name(ctypes.byref(a), buf, 64) == 0
        return buf.value.decode()
    lib.hirest_gemm_select_kernel(0)
    lib.hirest_gemm_debug_mode(0)
    Mv, Mp, Mt = 1024 * 257, 1024 * 256, 546 * 77
    problems = [(Mv, 4224, 1408, _lib.EPI_LNFOLD_BF16), (Mv, 1408, 1408, _lib.EPI_BIAS_RESID2_LNSTATS),
                (Mv, 6144, 1408, _lib.EPI_LNFOLD_GELU_BF16), (Mv, 1408, 6144, _lib.EPI_BIAS_RESID2_LNSTATS),
                # (HIREST_TOWER_F32_RESIDUAL: the fp32 residual array of rounds 1-3)
                (Mv, 1408, 1408, _lib.EPI_BIAS_RESID_LNSTATS_F32), (Mv, 1408, 6144, _lib.EPI_BIAS_RESID_LNSTATS_F32),
                (Mp, 1408, 640, _lib.EPI_PATCH_POS_F32), (1024, 1024, 1408, _lib.EPI_BIAS_F32),
                # text tower (546 prompts x 77 tokens, width 768): qkv, out_proj, c_fc, c_proj, projection
                (Mt, 2304, 768, _lib.EPI_BIAS_BF16), (Mt, 768, 768, _lib.EPI_BIAS_RESID_F32), (Mt, 3072, 768, _lib.EPI_BIAS_GELU_BF16),
                (Mt, 768, 3072, _lib.EPI_BIAS_RESID_F32), (546, 1024, 768, _lib.EPI_BIAS_F32)]
    dispatched = {name(*p) for p in problems}
    assert {"gemm_pq256<7>", "gemm_pq256<10>", "gemm_pq256<6>", "gemm_pq256<8>", "gemm_pq256<5>"} <= dispatched        # (round 4: the two-phase ping-pong kernel)
    rounds = sorted(d for d in os.listdir(os.path.join(REPO, "profiles")) if re.fullmatch(r"r\d\d", d))
    # the newest round that holds a traffic profile (a round's directory exists from its first committed measurement on; its
    # counters are collected on the final build)
    with_traffic = [r for r in rounds if os.path.isfile(os.path.join(REPO, "profiles", r, "pmc_traffic.json"))]
    if with_traffic[-1] != rounds[-1]:
        pytest.skip(f"profiles/{rounds[-1]} has no traffic profile yet (collected on the round's final build: tools/final_round.sh)")
    newest = os.path.join(REPO, "profiles", with_traffic[-1])
    src = open(os.path.join(REPO, "bench.py")).read()
    assert f'"{rounds[-1]}"' in src.split("PROFILE_ROUNDS")[1].split("\n")[0]      # bench.py reads the newest round first
    prof = json.load(open(os.path.join(newest, "pmc_traffic.json")))
    named = set()
    for k in prof["kernels"]:
        named |= set(re.findall(r"gemm_\w+<[^>]*>", k["kernel"]))
    sec1 = open(os.path.join(newest, "pmc_summary.md")).read().split("## 2.")[0]
    named |= set(re.findall(r"`(gemm_\w+<[^>`]*>)`", sec1))
    assert named, "no GEMM kernels found in the newest profile"
    assert named <= dispatched, f"profiled but no longer dispatched: {sorted(named - dispatched)}"
    # selection switches change the answer (the entry point mirrors the dispatch, it is not a constant table)
    lib.hirest_gemm_select_kernel(6)
    assert name(Mv, 1408, 6144, _lib.EPI_BIAS_RESID_LNSTATS_F32) == "gemm_p256<6, 64, false, 1>"
    for retired in (10, 17, 18, 20):                                      # the 4-wave kernel (round 2) and gemm_d2 (round 3)
        assert lib.hirest_gemm_select_kernel(retired) != 0
    lib.hirest_gemm_select_kernel(0)
