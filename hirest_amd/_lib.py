"""ctypes binding of libhirest_hip.so (include/hirest_hip.h).  No torch types cross the boundary:
device pointers are ``tensor.data_ptr()`` integers, the stream is the raw hipStream_t handle.

There is NO fallback: if the library is missing or fails to load, every op raises.  (The CPU
oracle under oracle/ is test infrastructure and is never imported from here.)
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libhirest_hip.so")
if os.environ.get("HIREST_LIB_VARIANT"):      # A/B builds of tools/build_variant.sh (hirest_amd/lib/libhirest_hip.<tag>.so): measurement only
    LIB_PATH = os.path.join(HERE, "lib", f"libhirest_hip.{os.environ['HIREST_LIB_VARIANT']}.so")

(EPI_BIAS_BF16, EPI_BIAS_GELU_BF16, EPI_BIAS_QGELU_BF16, EPI_BIAS_RESID_F32, EPI_BIAS_F32, EPI_PATCH_POS_F32,
 EPI_BIAS_RESID_LNSTATS_F32, EPI_LNFOLD_BF16, EPI_LNFOLD_GELU_BF16, EPI_BIAS_GELU_SPLIT2, EPI_BIAS_RESID2_LNSTATS) = range(11)

TOWER_NO_LNFOLD = 1
TOWER_NO_PRUNE = 2
TOWER_F32_RESIDUAL = 4
GEMM_REVERSE = 1
GEMM_X3 = 2
GEMM_X3_T128 = 4
ABI_VERSION = 4   # HIREST_ABI_VERSION of include/hirest_hip.h this binding mirrors

ERRORS = {-1: "HIREST_E_BADARG", -2: "HIREST_E_SHAPE (unsupported shape)", -3: "HIREST_E_WORKSPACE (workspace too small)"}


class GemmArgs(C.Structure):
    _fields_ = [("struct_size", C.c_uint64),
                ("A", C.c_void_p), ("lda", C.c_int64), ("W", C.c_void_p), ("ldw", C.c_int64),
                ("bias", C.c_void_p), ("out", C.c_void_p), ("ldo", C.c_int64),
                ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("epilogue", C.c_int32),
                ("pos", C.c_void_p), ("patches_per_frame", C.c_int32), ("aux0", C.c_void_p), ("aux1", C.c_void_p), ("flags", C.c_int32)]

    @classmethod
    def make(cls, *fields):
        """GemmArgs with struct_size filled in; `fields` are the members after it, in header order."""
        return cls(C.sizeof(cls), *fields)



class BlockWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("ln1_g", "ln1_b", "qkv_w", "qkv_b", "proj_w", "proj_b",
                                          "ln2_g", "ln2_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b",
                                          "qkv_wf", "qkv_bf", "qkv_s", "fc1_wf", "fc1_bf", "fc1_s")]


class VisionTower(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("image_size", "patch", "width", "heads", "head_dim", "mlp_dim",
                                         "layers", "embed_dim", "kpad", "act")] + \
               [("ln_eps", C.c_float)] + \
               [("patch_w", C.c_void_p), ("patch_b", C.c_void_p), ("cls", C.c_void_p), ("pos", C.c_void_p),
                ("blocks", C.POINTER(BlockWeights)),
                ("norm_g", C.c_void_p), ("norm_b", C.c_void_p), ("head_w", C.c_void_p), ("head_b", C.c_void_p),
                ("image_mean", C.c_void_p), ("image_std", C.c_void_p),
                ("ln_pre_g", C.c_void_p), ("ln_pre_b", C.c_void_p), ("out_all_tokens", C.c_int32)]


class TextTower(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("context", "vocab", "width", "heads", "layers", "embed_dim", "act")] + \
               [("ln_eps", C.c_float)] + \
               [("tok_emb", C.c_void_p), ("pos", C.c_void_p), ("blocks", C.POINTER(BlockWeights)),
                ("lnf_g", C.c_void_p), ("lnf_b", C.c_void_p), ("proj_w", C.c_void_p)]


class BlockWeightsF32(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("ln1_g", "ln1_b", "qkv_w", "qkv_b", "proj_w", "proj_b",
                                          "ln2_g", "ln2_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b")]


class VisionTowerF32(C.Structure):     # hirest_vision_tower_f32: the same fields with fp32 weight pointers
    _fields_ = [(n, t) if n != "blocks" else (n, C.POINTER(BlockWeightsF32)) for n, t in VisionTower._fields_]


class TextTowerF32(C.Structure):
    _fields_ = [(n, t) if n != "blocks" else (n, C.POINTER(BlockWeightsF32)) for n, t in TextTower._fields_]


class BlockWeightsX3(C.Structure):    # hirest_block_weights_x3: split (hi | lo) bf16 weights [out, 2 * in]
    _fields_ = [(n, C.c_void_p) for n in ("qkv_w2", "proj_w2", "fc1_w2", "fc2_w2")]


class VisionTowerX3(C.Structure):
    _fields_ = [("base", C.POINTER(VisionTowerF32)), ("blocks", C.POINTER(BlockWeightsX3))]


class CaptionLayer(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("qkv_w", "qkv_b", "so_w", "so_b", "so_ln_g", "so_ln_b", "cq_w", "cq_b",
                                          "co_w", "co_b", "co_ln_g", "co_ln_b", "ff1_w", "ff1_b", "ff2_w", "ff2_b", "ff_ln_g", "ff_ln_b")]


class CaptionDecoder(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("layers", "heads", "hidden", "inter", "vocab_padded", "max_pos")] + \
               [(n, C.c_void_p) for n in ("word_emb", "pos_emb", "emb_ln_g", "emb_ln_b")] + \
               [("layer", C.POINTER(CaptionLayer))] + \
               [(n, C.c_void_p) for n in ("tr_w", "tr_b", "tr_ln_g", "tr_ln_b", "lm_w", "lm_b", "lm_w2")]


class JointLayerX3(C.Structure):
    """hirest_joint_layer_x3 (include/hirest_hip.h)."""
    _fields_ = [(n, C.c_void_p) for n in ("qkv_w2", "qkv_b", "ao_w2", "ao_b", "ln1_g", "ln1_b", "fc1_w2", "fc1_b", "fc2_w2", "fc2_b", "ln2_g", "ln2_b")]


class JointEncoderX3(C.Structure):
    """hirest_joint_encoder_x3 (include/hirest_hip.h)."""
    _fields_ = [("struct_size", C.c_uint64)] + [(n, C.c_int32) for n in ("layers", "heads", "width", "mlp_dim", "in_dim", "max_pos")] + \
               [("ln_eps", C.c_float), ("attn_shift", C.c_float)] + \
               [(n, C.c_void_p) for n in ("emb_w2", "emb_b", "pos", "emb_ln_g", "emb_ln_b")] + [("layer", C.POINTER(JointLayerX3))]


class ColsumItem(C.Structure):
    """hirest_colsum_item (include/hirest_hip.h)."""
    _fields_ = [("x", C.c_void_p), ("row_weight", C.c_void_p), ("row_select", C.c_void_p), ("out", C.c_void_p), ("ldx", C.c_int64),
                ("R", C.c_int32), ("C", C.c_int32), ("select_value", C.c_int32), ("reserved", C.c_int32)]


COLSUM_GROUP_MAX = 40


class SplitItem(C.Structure):
    """hirest_split_item (include/hirest_hip.h)."""
    _fields_ = [("x", C.c_void_p), ("out", C.c_void_p), ("ldx", C.c_int64), ("ldo", C.c_int64), ("rows", C.c_int32), ("cols", C.c_int32),
                ("transposed", C.c_int32), ("reserved", C.c_int32)]


class TrainBlock(C.Structure):
    """hirest_train_block (include/hirest_hip.h): one post-LN encoder block in train mode, forward + what its backward needs."""
    _fields_ = [("struct_size", C.c_uint64)] + [(n, C.c_int32) for n in ("B", "T", "heads", "width", "mlp", "precision")] + \
               [("ln_eps", C.c_float), ("drop", C.c_float)] + [(n, C.c_uint32) for n in ("seed_attn", "seed_ao", "seed_out", "reserved")] + \
               [(n, C.c_void_p) for n in ("wqkv", "bqkv", "wo", "bo", "ln1_g", "ln1_b", "w1", "b1", "w2", "b2", "ln2_g", "ln2_b", "x",
                                          "qkv", "P", "cx", "a_pre", "aa", "hpre", "hh", "x_pre", "out", "x2", "out2",
                                          "wqkv2", "wo2", "w12", "w22", "wqkvT2", "woT2", "w1T2", "w2T2", "ws")] + [("ws_bytes", C.c_size_t)]


class TrainFusionBwd(C.Structure):
    """hirest_train_fusion_bwd (include/hirest_hip.h)."""
    _fields_ = [("struct_size", C.c_uint64)] + \
               [(n, C.c_int32) for n in ("B", "T", "E", "W", "vis_dim", "text_dim", "asr_dim", "boundary", "max_pos", "reserved")] + \
               [("drop", C.c_float), ("seed_emb", C.c_uint32)] + \
               [(n, C.c_void_p) for n in ("w_emb", "emb_ln_g", "t2_w", "asr1_w", "asr0_g", "norm_g",
                                          "x0", "f", "v", "tn", "tin", "a0", "asr2", "v0", "vis2", "t", "text", "mm32", "bm32", "n_valid", "dx",
                                          "g_emb_ln_g", "g_emb_ln_b", "g_pos", "g_w_emb", "g_b_emb", "g_mask", "g_bound", "g_t2_w", "g_t2_b", "g_t0_w", "g_t0_b",
                                          "g_asr1_w", "g_asr1_b", "g_asr0_g", "g_asr0_b", "g_norm_g", "g_norm_b", "g_vis_w", "g_vis_b", "g_text_w", "g_text_b")] + \
               [("items", C.POINTER(ColsumItem)), ("n_items", C.POINTER(C.c_int32)), ("max_items", C.c_int32), ("reserved2", C.c_int32),
                ("ws", C.c_void_p), ("ws_bytes", C.c_size_t), ("side_stream", C.c_void_p), ("side_ws", C.c_void_p), ("side_ws_bytes", C.c_size_t),
                ("scratch", C.c_void_p), ("scratch_bytes", C.c_size_t)]


class TrainBlockGrads(C.Structure):
    """hirest_train_block_grads (include/hirest_hip.h)."""
    _fields_ = [("struct_size", C.c_uint64)] + \
               [(n, C.c_void_p) for n in ("dout", "dx", "g_wqkv", "g_wo", "g_w1", "g_w2", "g_bqkv", "g_bo", "g_b1", "g_b2", "g_ln1_g", "g_ln1_b",
                                          "g_ln2_g", "g_ln2_b")] + \
               [("items", C.POINTER(ColsumItem)), ("n_items", C.POINTER(C.c_int32)), ("max_items", C.c_int32), ("reserved", C.c_int32),
                ("side_stream", C.c_void_p), ("side_ws", C.c_void_p), ("side_ws_bytes", C.c_size_t), ("scratch", C.c_void_p),
                ("scratch_bytes", C.c_size_t)]


class ProfRecord(C.Structure):
    _fields_ = [("kind", C.c_int32), ("tag", C.c_int32), ("d0", C.c_int64), ("d1", C.c_int64), ("d2", C.c_int64),
                ("ms", C.c_float)]


_SIGNATURES = {
    "hirest_profile_enable": (C.c_int, [C.c_int32]),
    "hirest_attention_debug_mode": (C.c_int, [C.c_int32]),
    "hirest_profile_collect": (C.c_int, [C.POINTER(ProfRecord), C.c_int32]),
    "hirest_abi_version": (C.c_int, []),
    "hirest_build_info": (C.c_char_p, []),
    "hirest_gemm_bf16": (C.c_int, [C.POINTER(GemmArgs), C.c_void_p]),
    "hirest_gemm_select_kernel": (C.c_int, [C.c_int32]),
    "hirest_gemm_debug_mode": (C.c_int, [C.c_int32]),
    "hirest_gemm_dispatch_name": (C.c_int, [C.POINTER(GemmArgs), C.c_char_p, C.c_int32]),
    "hirest_attention_select_kernel": (C.c_int, [C.c_int32]),
    "hirest_attention_set_skew": (C.c_int, [C.c_int32]),
    "hirest_attention_set_mapping": (C.c_int, [C.c_int32]),
    "hirest_attention_set_pace": (C.c_int, [C.c_int32]),
    "hirest_attention_x3_select_waves": (C.c_int, [C.c_int32]),
    "hirest_attention_x3_debug_trace": (C.c_int, [C.c_void_p]),
    "hirest_attention_set_stagger": (C.c_int, [C.c_int32]),
    "hirest_attention_debug_trace_read": (C.c_int, [C.c_void_p, C.c_int32]),
    "hirest_layernorm": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float,
                                   C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "hirest_attention_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                        C.c_float, C.c_int32, C.c_void_p]),
    "hirest_attention_bf16_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                             C.c_float, C.c_int32, C.c_int32, C.c_void_p]),
    "hirest_fold_layernorm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                        C.c_int32, C.c_void_p]),
    "hirest_patchify": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_int32, C.c_void_p]),
    "hirest_rowstats_bf16": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_float, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "hirest_rowstats_split_bf16": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int32, C.c_int32, C.c_void_p,
                                             C.c_void_p]),
    "hirest_combine_hi_lo_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p]),
    "hirest_ln_stats_finalize": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_float, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "hirest_write_cls_rows": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                        C.c_int32, C.c_void_p]),
    "hirest_embed_tokens": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                      C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "hirest_f32_to_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "hirest_pool_l2norm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "hirest_pool_l2norm_varlen": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "hirest_similarity_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "hirest_topk_workspace_bytes": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32]),
    "hirest_topk_f32_ws": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_int64, C.c_void_p]),
    "hirest_topk_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                  C.c_void_p]),
    "hirest_gemm_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                  C.c_int32, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "hirest_attention_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float,
                                       C.c_float, C.c_void_p]),
    "hirest_attention_f32_qkv": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_int32,
                                           C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_void_p]),
    "hirest_gemm_f32_select_kernel": (C.c_int, [C.c_int32]),
    "hirest_gemm_f32_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    "hirest_gemm_f32_layouts_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    "hirest_gemm_f32_layouts": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64,
                                          C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p]),
    "hirest_gemm_f32_ws": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                     C.c_int32, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_size_t,
                                     C.c_void_p]),
    "hirest_attention_f32_decode": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p,
                                              C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_float,
                                              C.c_float, C.c_float, C.c_void_p]),
    "hirest_gemm_f32_ln": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float,
                                     C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                     C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "hirest_caption_step_workspace_bytes": (C.c_size_t, [C.POINTER(CaptionDecoder), C.c_int32]),
    "hirest_beam_backtrack": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "hirest_beam_advance": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p]),
    "hirest_caption_decode_step": (C.c_int, [C.POINTER(CaptionDecoder), C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                             C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int32,
                                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "hirest_caption_decode_logits": (C.c_int, [C.POINTER(CaptionDecoder), C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                               C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int32,
                                               C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "hirest_caption_select": (C.c_int, [C.c_int32]),
    "hirest_gemm_f32_ln_colmax": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int64, C.c_void_p,
                                            C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "hirest_gemm_f32_rows_ln_mode": (C.c_int, [C.c_int32]),
    "hirest_gemm_f32_ring_mode": (C.c_int, [C.c_int32]),
    "hirest_gemm_f32_rows_preferred": (C.c_int, [C.c_int32]),
    "hirest_gemm_f32_rows_colmax": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                              C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "hirest_caption_beam_step": (C.c_int, [C.POINTER(CaptionDecoder), C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                           C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int32, C.c_void_p,
                                           C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]),
    "hirest_caption_beam_tail_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    "hirest_caption_beam_tail": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                           C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "hirest_attention_f32_varlen": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float,
                                              C.c_float, C.c_void_p]),
    "hirest_log_softmax_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p]),
    "hirest_joint_time_grid_f32": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "hirest_joint_time_features": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                             C.c_void_p]),
    "hirest_joint_base": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                    C.c_void_p]),
    "hirest_joint_mask_add": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                        C.c_int32, C.c_void_p]),
    "hirest_transpose_pad_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]),
    "hirest_weighted_colsum_grouped_f32": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p]),
    "hirest_train_fusion_backward_scratch_bytes": (C.c_size_t, [C.POINTER(TrainFusionBwd)]),
    "hirest_train_fusion_backward": (C.c_int, [C.POINTER(TrainFusionBwd), C.c_void_p]),
    "hirest_train_block_forward_scratch_bytes": (C.c_size_t, [C.POINTER(TrainBlock)]),
    "hirest_train_block_forward": (C.c_int, [C.POINTER(TrainBlock), C.c_void_p, C.c_size_t, C.c_void_p]),
    "hirest_train_block_backward_scratch_bytes": (C.c_size_t, [C.POINTER(TrainBlock)]),
    "hirest_train_block_backward": (C.c_int, [C.POINTER(TrainBlock), C.POINTER(TrainBlockGrads), C.c_void_p]),
    "hirest_weighted_colsum_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "hirest_scale_by_device_scalar_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "hirest_act_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]),
    "hirest_act_bwd_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]),
    "hirest_dropout_add_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_uint32, C.c_void_p]),
    "hirest_layernorm_bwd_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "hirest_attention_train_select": (C.c_int, [C.c_int32]),
    "hirest_gemm_f32_strided": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64,
                                          C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_void_p]),
    "hirest_attention_train_fwd_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float,
                                                 C.c_float, C.c_float, C.c_uint32, C.c_void_p]),
    "hirest_attention_train_bwd_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                                 C.c_int32, C.c_float, C.c_float, C.c_uint32, C.c_void_p]),
    "hirest_attention_train_fwd_qkv_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                                     C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float,
                                                     C.c_float, C.c_uint32, C.c_void_p]),
    "hirest_attention_train_bwd_qkv_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64,
                                                     C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32,
                                                     C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_uint32, C.c_void_p]),
    "hirest_embedding_fwd_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p]),
    "hirest_embedding_pos_fwd_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]),
    "hirest_embedding_bwd_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]),
    "hirest_ce_rows_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hirest_bce_masked_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hirest_joint_base_bwd_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "hirest_l2norm_bwd_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "hirest_ce_masked_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hirest_heads_bwd_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hirest_linear_heads": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p]),
    "hirest_masked_argmax": (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "hirest_segmentation_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_double, C.c_void_p,
                                           C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "hirest_preprocess_plan_bytes": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32]),
    "hirest_preprocess_plan": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int64]),
    "hirest_preprocess_workspace_bytes": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "hirest_preprocess_u8": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                       C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "hirest_interval_iou_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "hirest_step_bound_pr": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_double, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p]),
    "hirest_frame_to_timestamp": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]),
    "hirest_timestamp_to_frame": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]),
    "hirest_preprocess_moment_bounds": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                                  C.c_int32, C.c_void_p]),
    "hirest_vision_workspace_bytes": (C.c_size_t, [C.POINTER(VisionTower), C.c_int32]),
    "hirest_vision_forward": (C.c_int, [C.POINTER(VisionTower), C.c_void_p, C.c_int32, C.c_int32, C.c_void_p,
                                        C.c_void_p, C.c_size_t, C.c_int32, C.c_void_p]),
    "hirest_vision_guard_offset": (C.c_size_t, [C.POINTER(VisionTower), C.c_int32]),
    "hirest_vision_workspace_bytes_f32": (C.c_size_t, [C.POINTER(VisionTowerF32), C.c_int32]),
    "hirest_vision_embed_f32": (C.c_int, [C.POINTER(VisionTowerF32), C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hirest_joint_encoder_x3_workspace_bytes": (C.c_size_t, [C.POINTER(JointEncoderX3), C.c_int32, C.c_int32]),
    "hirest_joint_encoder_x3_forward": (C.c_int, [C.POINTER(JointEncoderX3), C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t,
                                                  C.c_void_p]),
    "hirest_layernorm_f32_split2": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int64,
                                              C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p]),
    "hirest_split2_bf16": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_void_p]),
    "hirest_split2_grouped_bf16": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p]),
    "hirest_split2_transposed_bf16": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p]),
    "hirest_layernorm_split2": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int64, C.c_int32,
                                          C.c_int32, C.c_void_p]),
    "hirest_attention_x3_qkv": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                          C.c_int32, C.c_int32, C.c_float, C.c_void_p]),
    "hirest_attention_x3_qkv_split2": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_int32,
                                                 C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_void_p]),
    "hirest_vision_x3_select_attention": (C.c_int, [C.c_int32]),
    "hirest_vision_workspace_bytes_x3": (C.c_size_t, [C.POINTER(VisionTowerX3), C.c_int32]),
    "hirest_vision_forward_x3": (C.c_int, [C.POINTER(VisionTowerX3), C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t,
                                           C.c_void_p]),
    "hirest_vision_forward_f32": (C.c_int, [C.POINTER(VisionTowerF32), C.c_void_p, C.c_int32, C.c_int32, C.c_void_p,
                                            C.c_void_p, C.c_size_t, C.c_void_p]),
    "hirest_text_workspace_bytes_f32": (C.c_size_t, [C.POINTER(TextTowerF32), C.c_int32]),
    "hirest_text_forward_f32": (C.c_int, [C.POINTER(TextTowerF32), C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                          C.c_size_t, C.c_void_p]),
    "hirest_text_workspace_bytes": (C.c_size_t, [C.POINTER(TextTower), C.c_int32]),
    "hirest_text_forward": (C.c_int, [C.POINTER(TextTower), C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                      C.c_size_t, C.c_void_p]),
}

EXPORTS = tuple(_SIGNATURES)
_lib = None


def load():
    """Load the library once; raise (never fall back) when it is unavailable."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: build it with `python -m hirest_amd.build` "
                           "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    # the version check comes first: a stale .so then fails with "rebuild", not with an AttributeError on a symbol it predates
    lib.hirest_abi_version.restype, lib.hirest_abi_version.argtypes = C.c_int, []
    if lib.hirest_abi_version() != ABI_VERSION:
        raise RuntimeError(f"libhirest_hip.so ABI version {lib.hirest_abi_version()} != binding {ABI_VERSION}: rebuild with "
                           "`python -m hirest_amd.build --force`")
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype, fn.argtypes = res, args
    if os.environ.get("HIREST_ATTENTION_KERNEL"):      # A/B timing of the bf16 attention forms without touching the caller
        check(lib.hirest_attention_select_kernel(int(os.environ["HIREST_ATTENTION_KERNEL"])), "HIREST_ATTENTION_KERNEL")
    _lib = lib
    return lib


def check(code: int, what: str):
    if code == 0:
        return
    if code < 0:
        raise RuntimeError(f"{what}: {ERRORS.get(code, code)}")
    raise RuntimeError(f"{what}: HIP error {code}")
