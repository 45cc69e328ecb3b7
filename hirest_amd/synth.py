"""Deterministic synthetic weights / inputs (no pretrained checkpoints exist offline).

Every value is a pure function of (seed, tensor name, flat index): a splitmix64 hash of
the index, top 24 bits -> uniform in [-1, 1) -> scaled.  Only integer ops and exact
float64 multiplies are used, so the generated tensors are bit-identical on every
machine / numpy / torch build (``torch.manual_seed`` gives no such guarantee).  The
golden-vector script, the CPU oracle, the parity tests and ``bench.py`` all draw their
weights from here, so "same inputs" is true by construction on both sides of a
comparison.

The state-dict *key names and shapes* are the reference's checkpoint schema
(``eva_clip_psz14.pt``: /root/reference/EVA_clip/eva_model.py:177-315,
vit_model.py:248-310; OpenAI CLIP: EVA_clip/model.py:216-331), so the same dict loads
into the reference modules via ``load_state_dict`` and into this package's towers.
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import numpy as np
import torch

_M64 = (1 << 64) - 1


def _fnv1a64(s: str) -> int:
    h = 0xCBF29CE484222325
    for b in s.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & _M64
    return h


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def uniform_pm1(name: str, n: int, seed: int) -> np.ndarray:
    """n float64 values, uniform on the 2^-23 grid in [-1, 1)."""
    base = np.uint64((_fnv1a64(name) ^ ((seed * 0x9E3779B97F4A7C15) & _M64)) & _M64)
    out = np.empty(n, dtype=np.float64)
    step = 1 << 22
    for s in range(0, n, step):
        e = min(n, s + step)
        with np.errstate(over="ignore"):
            idx = np.arange(s, e, dtype=np.uint64) + base
        z = _splitmix64(idx)
        out[s:e] = (z >> np.uint64(40)).astype(np.float64) * (2.0 / (1 << 24)) - 1.0
    return out


def tensor(name: str, shape, std: float, seed: int, mean: float = 0.0) -> torch.Tensor:
    """fp32 tensor with the given std (uniform distribution of matching variance).  Same values as
    ``(uniform_pm1(...) * (std * sqrt(3)) + mean).astype(float32)``, produced chunk by chunk straight into the fp32
    result (a 1 B-parameter checkpoint never exists in float64)."""
    n = int(np.prod(shape)) if len(shape) else 1
    base = np.uint64((_fnv1a64(name) ^ ((seed * 0x9E3779B97F4A7C15) & _M64)) & _M64)
    v = np.empty(n, dtype=np.float32)
    scale = std * math.sqrt(3.0)
    step = 1 << 20
    for s in range(0, n, step):
        e = min(n, s + step)
        with np.errstate(over="ignore"):
            idx = np.arange(s, e, dtype=np.uint64) + base
        u = (_splitmix64(idx) >> np.uint64(40)).astype(np.float64) * (2.0 / (1 << 24)) - 1.0
        v[s:e] = u * scale + mean
    return torch.from_numpy(v.reshape(tuple(shape)))


def frames(name: str, shape, seed: int) -> torch.Tensor:
    """Synthetic already-normalised frames: unit-variance values (N(0,1)-like)."""
    return tensor(name, shape, 1.0, seed)


def rgb_frames(name: str, shape, seed: int) -> np.ndarray:
    """Synthetic decoded video frames, uint8 [..., H, W, 3]: 8x8-pixel flat blocks (hard edges, so the bicubic
    resampler's negative lobes overshoot and the 0/255 clamps are exercised) with per-pixel noise on half of them."""
    shape = tuple(int(v) for v in shape)
    *lead, h, w, c = shape
    nb = int(np.prod(lead)) if lead else 1
    fine = ((uniform_pm1(name + ".fine", nb * h * w * c, seed) + 1.0) * 128.0).astype(np.int64).clip(0, 255)
    fine = fine.reshape(nb, h, w, c)
    bh, bw = (h + 7) // 8, (w + 7) // 8
    coarse = ((uniform_pm1(name + ".coarse", nb * bh * bw * c, seed) + 1.0) * 128.0).astype(np.int64).clip(0, 255)
    coarse = np.repeat(np.repeat(coarse.reshape(nb, bh, bw, c), 8, axis=1), 8, axis=2)[:, :h, :w]
    pick = uniform_pm1(name + ".pick", nb * bh * bw, seed).reshape(nb, bh, bw, 1) > 0.0
    pick = np.repeat(np.repeat(pick, 8, axis=1), 8, axis=2)[:, :h, :w]
    out = np.where(pick, coarse, (coarse + fine) // 2).astype(np.uint8)
    return out.reshape(shape)


def tokens(name: str, batch: int, seed: int, context_length: int = 77,
           vocab_size: int = 49408) -> torch.Tensor:
    """Synthetic CLIP token rows: [SOT] ids... [EOT] 0-padding (clip.py:196-232 layout).

    EOT (vocab-1) is the largest id in each row, which is what ``text.argmax(-1)``
    (eva_model.py:243) relies on.
    """
    sot, eot = vocab_size - 2, vocab_size - 1
    u = uniform_pm1(name, batch * context_length, seed).reshape(batch, context_length)
    lens = 3 + ((uniform_pm1(name + ".len", batch, seed) + 1.0) * 0.5 * 20).astype(np.int64)
    ids = ((u + 1.0) * 0.5 * (vocab_size - 1000)).astype(np.int64) + 300
    out = np.zeros((batch, context_length), dtype=np.int64)
    for b in range(batch):
        n = int(lens[b])
        out[b, 0] = sot
        out[b, 1:1 + n] = ids[b, :n]
        out[b, 1 + n] = eot
    return torch.from_numpy(out)


# ----------------------------------------------------------------------------------
# EVA-CLIP (vision: BEiT-style ViT; text: open_clip-style transformer)
# ----------------------------------------------------------------------------------

EVA_CLIP_G_14 = {  # /root/reference/EVA_clip/model_configs/EVA_CLIP_g_14.json
    "embed_dim": 1024,
    "vision_cfg": {"image_size": 224, "layers": 40, "width": 1408, "head_width": 88,
                   "mlp_ratio": 4.3637, "patch_size": 14, "drop_path_rate": 0.4},
    "text_cfg": {"context_length": 77, "vocab_size": 49408, "width": 768, "heads": 12,
                 "layers": 12},
}

# A small config with the awkward dimensions of the real one (head_dim 88, 257 tokens,
# mlp_ratio that truncates) for fast CPU parity checks.  Widths are multiples of 64 (the
# GEMM's K granule): 704 = 8 heads x 88.
EVA_CLIP_TINY = {
    "embed_dim": 64,
    "vision_cfg": {"image_size": 224, "layers": 2, "width": 704, "head_width": 88,
                   "mlp_ratio": 4.3637, "patch_size": 14, "drop_path_rate": 0.0},
    "text_cfg": {"context_length": 77, "vocab_size": 49408, "width": 128, "heads": 2,
                 "layers": 2},
}


def eva_vision_shapes(cfg: dict) -> Dict[str, Tuple[int, ...]]:
    v = cfg["vision_cfg"]
    D, P, L = v["width"], v["patch_size"], v["layers"]
    N = (v["image_size"] // P) ** 2 + 1
    Dm = int(D * v["mlp_ratio"])  # vit_model.py:166
    E = cfg["embed_dim"]
    s = {"visual.cls_token": (1, 1, D), "visual.pos_embed": (1, N, D),
         "visual.patch_embed.proj.weight": (D, 3, P, P), "visual.patch_embed.proj.bias": (D,)}
    for i in range(L):
        p = f"visual.blocks.{i}."
        s.update({p + "norm1.weight": (D,), p + "norm1.bias": (D,),
                  p + "attn.q_bias": (D,), p + "attn.v_bias": (D,),
                  p + "attn.qkv.weight": (3 * D, D),
                  p + "attn.proj.weight": (D, D), p + "attn.proj.bias": (D,),
                  p + "norm2.weight": (D,), p + "norm2.bias": (D,),
                  p + "mlp.fc1.weight": (Dm, D), p + "mlp.fc1.bias": (Dm,),
                  p + "mlp.fc2.weight": (D, Dm), p + "mlp.fc2.bias": (D,)})
    s.update({"visual.norm.weight": (D,), "visual.norm.bias": (D,),
              "visual.head.weight": (E, D), "visual.head.bias": (E,)})
    return s


def eva_text_shapes(cfg: dict) -> Dict[str, Tuple[int, ...]]:
    t = cfg["text_cfg"]
    D, L, E = t["width"], t["layers"], cfg["embed_dim"]
    s = {"text.token_embedding.weight": (t["vocab_size"], D),
         "text.positional_embedding": (t["context_length"], D)}
    for i in range(L):
        p = f"text.transformer.resblocks.{i}."
        s.update({p + "ln_1.weight": (D,), p + "ln_1.bias": (D,),
                  p + "attn.in_proj_weight": (3 * D, D), p + "attn.in_proj_bias": (3 * D,),
                  p + "attn.out_proj.weight": (D, D), p + "attn.out_proj.bias": (D,),
                  p + "ln_2.weight": (D,), p + "ln_2.bias": (D,),
                  p + "mlp.c_fc.weight": (4 * D, D), p + "mlp.c_fc.bias": (4 * D,),
                  p + "mlp.c_proj.weight": (D, 4 * D), p + "mlp.c_proj.bias": (D,)})
    s.update({"text.ln_final.weight": (D,), "text.ln_final.bias": (D,),
              "text.text_projection": (D, E), "text.logit_scale": ()})
    return s


def _layer_index(name: str) -> int:
    parts = name.split(".")
    for a, b in zip(parts, parts[1:]):
        if a in ("blocks", "resblocks", "layer") and b.isdigit():
            return int(b)
    return 0


def _init_rule(name: str, shape) -> Tuple[float, float]:
    """(std, mean) per tensor.  Scales follow the reference's own initialisers
    (vit_model.py:291-318 trunc_normal .02 with proj/fc2 / sqrt(2*layer);
    eva_model.py:206-222) but biases and LayerNorm affines are made non-trivial so
    that every bias / affine code path is exercised by the parity tests."""
    leaf = name.split(".")[-1]
    if name.endswith("logit_scale"):
        return 0.0, math.log(1 / 0.07)
    if "norm" in name or ".ln_" in name or "ln_final" in name or "ln_pre" in name or "ln_post" in name \
            or "LayerNorm" in name:
        return (0.1, 1.0) if leaf == "weight" else (0.05, 0.0)
    if leaf in ("bias", "q_bias", "v_bias", "in_proj_bias"):
        return 0.02, 0.0
    if name.endswith("token_embedding.weight"):
        return 0.02, 0.0
    if name.endswith("positional_embedding") or name.endswith("pos_embed") or name.endswith("cls_token") \
            or name.endswith("class_embedding"):
        return 0.02, 0.0
    if name.endswith("qkv.weight") or name.endswith("in_proj_weight"):
        return 0.04, 0.0
    if name.endswith("attn.proj.weight") or name.endswith("fc2.weight") \
            or name.endswith("out_proj.weight") or name.endswith("c_proj.weight"):
        return 0.02 / math.sqrt(2.0 * (_layer_index(name) + 1)), 0.0
    if name.endswith("text_projection") or name.endswith("visual.proj"):
        return shape[0] ** -0.5, 0.0
    return 0.02, 0.0


def state_dict(shapes: Dict[str, Tuple[int, ...]], seed: int) -> Dict[str, torch.Tensor]:
    # tensors are independent pure functions of (name, seed): generate them on a few threads (numpy releases the GIL)
    import os
    from concurrent.futures import ThreadPoolExecutor

    def make(item):
        name, shape = item
        std, mean = _init_rule(name, shape)
        return name, tensor(name, shape, std, seed, mean)
    items = list(shapes.items())
    with ThreadPoolExecutor(max_workers=max(1, min(16, os.cpu_count() or 1))) as ex:
        out = dict(ex.map(make, items))
    # the decoder's input embedding and its LM-head matrix are ONE tied parameter in the reference
    # (module_decoder.py:171-176,284-285): a checkpoint carries the same values under both names
    tied_a = "clip4cap_model.decoder.embeddings.word_embeddings.weight"
    tied_b = "clip4cap_model.decoder.classifier.cls.predictions.decoder.weight"
    if tied_a in out and tied_b in out:
        out[tied_a] = out[tied_b]
    return out


def eva_clip_state_dict(cfg: dict, seed: int, towers=("visual", "text")) -> Dict[str, torch.Tensor]:
    shapes = {}
    if "visual" in towers:
        shapes.update(eva_vision_shapes(cfg))
    if "text" in towers:
        shapes.update(eva_text_shapes(cfg))
    return state_dict(shapes, seed)


# ----------------------------------------------------------------------------------
# OpenAI CLIP ViT-B/32 as vendored in the reference (EVA_clip/model.py)
# ----------------------------------------------------------------------------------

OPENAI_VIT_B32 = {"embed_dim": 512, "image_resolution": 224, "vision_layers": 12, "vision_width": 768,
                  "vision_patch_size": 32, "context_length": 77, "vocab_size": 49408,
                  "transformer_width": 512, "transformer_heads": 8, "transformer_layers": 12}

OPENAI_VIT_TINY = {"embed_dim": 64, "image_resolution": 224, "vision_layers": 2, "vision_width": 128,
                   "vision_patch_size": 32, "context_length": 77, "vocab_size": 49408,
                   "transformer_width": 128, "transformer_heads": 2, "transformer_layers": 2}


def openai_clip_shapes(c: dict) -> Dict[str, Tuple[int, ...]]:
    W, P, E = c["vision_width"], c["vision_patch_size"], c["embed_dim"]
    G = c["image_resolution"] // P
    s = {"visual.conv1.weight": (W, 3, P, P), "visual.class_embedding": (W,),
         "visual.positional_embedding": (G * G + 1, W),
         "visual.ln_pre.weight": (W,), "visual.ln_pre.bias": (W,)}

    def blocks(prefix, D, L):
        for i in range(L):
            p = f"{prefix}.resblocks.{i}."
            s.update({p + "attn.in_proj_weight": (3 * D, D), p + "attn.in_proj_bias": (3 * D,),
                      p + "attn.out_proj.weight": (D, D), p + "attn.out_proj.bias": (D,),
                      p + "ln_1.weight": (D,), p + "ln_1.bias": (D,),
                      p + "mlp.c_fc.weight": (4 * D, D), p + "mlp.c_fc.bias": (4 * D,),
                      p + "mlp.c_proj.weight": (D, 4 * D), p + "mlp.c_proj.bias": (D,),
                      p + "ln_2.weight": (D,), p + "ln_2.bias": (D,)})

    blocks("visual.transformer", W, c["vision_layers"])
    s.update({"visual.ln_post.weight": (W,), "visual.ln_post.bias": (W,), "visual.proj": (W, E)})
    T = c["transformer_width"]
    blocks("transformer", T, c["transformer_layers"])
    s.update({"token_embedding.weight": (c["vocab_size"], T),
              "positional_embedding": (c["context_length"], T),
              "ln_final.weight": (T,), "ln_final.bias": (T,),
              "text_projection": (T, E), "logit_scale": ()})
    return s


def openai_clip_state_dict(c: dict, seed: int) -> Dict[str, torch.Tensor]:
    return state_dict(openai_clip_shapes(c), seed)


# ----------------------------------------------------------------------------------
# Joint model (MomentModel minus the frozen EVA-CLIP): schema = the reference's own state-dict keys
# (tests/golden/joint_schema.json, dumped from the real module)
# ----------------------------------------------------------------------------------

def joint_state_dict(shapes: Dict[str, Tuple[int, ...]], seed: int) -> Dict[str, torch.Tensor]:
    out = {}
    for name, shape in shapes.items():
        leaf = name.split(".")[-1]
        if "LayerNorm" in name or "visual_norm2d" in name or name.startswith("asr_enc_layer.0"):
            std, mean = (0.1, 1.0) if leaf == "weight" else (0.05, 0.0)
        elif leaf == "bias":
            std, mean = 0.02, 0.0
        elif "predictor" in name:
            std, mean = 0.05, 0.0
        elif name.endswith("query.weight") or name.endswith("key.weight"):
            std, mean = 0.06, 0.0          # non-trivial attention logits
        elif "embed" in name.lower():
            std, mean = 0.05, 0.0
        else:
            std, mean = 0.03, 0.0
        out[name] = tensor(name, shape, std, seed, mean)
    # the decoder's input embedding and its LM-head matrix are ONE tied parameter in the reference
    # (module_decoder.py:171-176,284-285): a checkpoint carries the same values under both names
    tied_a = "clip4cap_model.decoder.embeddings.word_embeddings.weight"
    tied_b = "clip4cap_model.decoder.classifier.cls.predictions.decoder.weight"
    if tied_a in out and tied_b in out:
        out[tied_a] = out[tied_b]
    return out


# ----------------------------------------------------------------------------------
# ASR sentence encoder: sentence-transformers/all-MiniLM-L6-v2 = a 6-layer BERT (extraction/whisper_ASR/extract_ASR_embedding.py:14)
# in the Hugging Face BertModel checkpoint schema (the file sentence-transformers loads with AutoModel)
# ----------------------------------------------------------------------------------

MINILM_L6 = {"vocab_size": 30522, "hidden_size": 384, "num_hidden_layers": 6, "num_attention_heads": 12,
             "intermediate_size": 1536, "max_position_embeddings": 512, "type_vocab_size": 2, "layer_norm_eps": 1e-12}
MINILM_TINY = {"vocab_size": 700, "hidden_size": 64, "num_hidden_layers": 2, "num_attention_heads": 2,
               "intermediate_size": 128, "max_position_embeddings": 64, "type_vocab_size": 2, "layer_norm_eps": 1e-12}


def bert_shapes(c: dict) -> Dict[str, Tuple[int, ...]]:
    D, I = c["hidden_size"], c["intermediate_size"]
    s = {"embeddings.word_embeddings.weight": (c["vocab_size"], D),
         "embeddings.position_embeddings.weight": (c["max_position_embeddings"], D),
         "embeddings.token_type_embeddings.weight": (c["type_vocab_size"], D),
         "embeddings.LayerNorm.weight": (D,), "embeddings.LayerNorm.bias": (D,)}
    for i in range(c["num_hidden_layers"]):
        p = f"encoder.layer.{i}."
        for n in ("attention.self.query", "attention.self.key", "attention.self.value", "attention.output.dense"):
            s[p + n + ".weight"] = (D, D)
            s[p + n + ".bias"] = (D,)
        s.update({p + "attention.output.LayerNorm.weight": (D,), p + "attention.output.LayerNorm.bias": (D,),
                  p + "intermediate.dense.weight": (I, D), p + "intermediate.dense.bias": (I,),
                  p + "output.dense.weight": (D, I), p + "output.dense.bias": (D,),
                  p + "output.LayerNorm.weight": (D,), p + "output.LayerNorm.bias": (D,)})
    s.update({"pooler.dense.weight": (D, D), "pooler.dense.bias": (D,)})   # in the checkpoint, unused by mean pooling
    return s


def bert_state_dict(c: dict, seed: int) -> Dict[str, torch.Tensor]:
    return joint_state_dict(bert_shapes(c), seed)


def sentence_ids(name: str, n: int, seed: int, vocab_size: int, min_len: int = 2, max_len: int = 40,
                 cls_id: int = 101, sep_id: int = 102) -> list:
    """n ragged rows [CLS] w.. [SEP] with lengths spread over [min_len, max_len] (min_len 2 = an empty subtitle)"""
    u = (uniform_pm1(name + ".len", n, seed) + 1.0) * 0.5
    rows = []
    for r in range(n):
        L = min_len + int(u[r] * (max_len - min_len + 1))
        L = min(max(L, 2), max_len)
        body = ((uniform_pm1(f"{name}.row{r}", L - 2, seed) + 1.0) * 0.5 * (vocab_size - 200)).astype(np.int64) + 200 if L > 2 \
            else np.zeros(0, np.int64)
        rows.append([cls_id] + [int(x) for x in body] + [sep_id])
    return rows


# ----------------------------------------------------------------------------------
# Seeded input builders of the joint-model / evaluation / timeline fixtures.  tests/golden/make_golden.py feeds these to the
# real reference; the GPU tests, tools/ and bench.py rebuild the same inputs from here (nothing in them touches the reference).
# ----------------------------------------------------------------------------------

def joint_inputs(name, B, T, seed):
    """C4-style synthetic batch: L2-normalised frame features, sparse ASR features, ragged lengths."""
    vis = tensor(f"{name}.vis", (B, T, 1024), 1.0, seed)
    vis = vis / vis.norm(dim=-1, keepdim=True)
    asr = tensor(f"{name}.asr", (B, T, 384), 0.05, seed)
    gaps = uniform_pm1(f"{name}.gap", B * T, seed).reshape(B, T) > 0.2      # ~40 % all-zero rows
    asr = asr * torch.from_numpy(~gaps).float()[..., None]
    text = tensor(f"{name}.text", (B, 1024), 1.0, seed)
    lens = [T - (b * T) // (3 * B) for b in range(B)]                             # ragged
    vis_mask = torch.zeros(B, T, dtype=torch.long)
    for b, n in enumerate(lens):
        vis_mask[b, :n] = 1
        vis[b, n:] = 0
        asr[b, n:] = 0
    bounds = torch.tensor([[int(0.1 * n), int(0.8 * n)] for n in lens], dtype=torch.long)
    moment_mask = torch.zeros(B, T, dtype=torch.long)
    for b in range(B):
        moment_mask[b, bounds[b, 0]:bounds[b, 1] + 1] = 1
    return vis, asr, text, vis_mask, moment_mask, bounds


TRAIN_CASES = {"a": (3, 64), "b": (2, 300)}
# step-captioning goldens (tests/golden/caption_predictions.json): case -> (B, T, beams, moment lengths)
CAPTION_CASES = {"a": (3, 64, 3, [7, 20, 37]), "b": (2, 300, 5, [7, 20]),
                 "c3": (5, 300, 3, [15] * 5), "c5": (5, 300, 5, [15] * 5),
                 # the reference's default evaluation batch (args.py:27 --eval_batch_size 32): 96 / 160 beam rows per word, moments
                 # shorter than, equal to and longer than max_frames mixed in one batch
                 "d3": (32, 300, 3, [[15, 7, 20, 37, 12, 25, 3, 18][b % 8] for b in range(32)]),
                 "d5": (32, 300, 5, [[15, 7, 20, 37, 12, 25, 3, 18][b % 8] for b in range(32)])}


def train_targets(name, B, T, seed, bounds):
    """start / end targets inside the moment, a previous-boundary mask and a segmentation target (hirest_dataset.py:409-531 keys)."""
    u = (uniform_pm1(f"{name}.tgt", 4 * B, seed).reshape(4, B) + 1.0) * 0.5
    lo, hi = bounds[:, 0].numpy(), bounds[:, 1].numpy()
    st = (lo + u[0] * (hi - lo)).astype(np.int64)
    et = np.maximum(st, (lo + u[1] * (hi - lo)).astype(np.int64))
    seg = (lo + u[2] * (hi - lo)).astype(np.int64)
    prev = torch.zeros(B, T, dtype=torch.long)
    for b in range(B):
        prev[b, int(lo[b])] = 1
    return torch.from_numpy(st), torch.from_numpy(et), torch.from_numpy(seg), prev


def caption_targets(name, B, max_words, seed):
    """Teacher-forcing triples in the reference's target_text 9-tuple layout (fields 5, 6, 7: decoder input ids, decoder mask,
    output ids with -1 on the padding): [CLS] w1 .. wn  /  w1 .. wn [SEP]."""
    u = (uniform_pm1(f"{name}.cap", B * (max_words + 1), seed).reshape(B, max_words + 1) + 1.0) * 0.5
    out = []
    for b in range(B):
        n = 3 + int(u[b, 0] * (max_words - 8))
        words = (1000 + (u[b, 1:1 + n] * 29000)).astype(np.int64).tolist()
        inp = [101] + words + [0] * (max_words - 1 - n)
        mask = [1] * (n + 1) + [0] * (max_words - 1 - n)
        outp = words + [102] + [-1] * (max_words - 1 - n)
        out.append((None, None, None, None, None, inp, mask, outp, None))
    return out


def moment_eval_inputs():
    """Seeded synthetic gt / predictions in evaluate.py's JSON layouts (shared with tests/test_evaluation.py)."""
    cats = ["Food", "Hobbies", "Home"]
    u = uniform_pm1("moment_eval", 20000, 17)
    it = iter(((u + 1.0) * 0.5).tolist())
    rnd = lambda lo, hi: lo + (hi - lo) * next(it)
    prompt_to_cat, video_to_cat = {}, {}
    mr_gt, mr_pred, sb_gt, sb_pred = {}, {}, {}, {}
    for pi in range(24):
        prompt = f"prompt {pi}"
        prompt_to_cat[prompt] = cats[pi % 3]
        mr_gt[prompt], mr_pred[prompt] = {}, {}
        for vi in range(1 + pi % 3):
            video = f"vid_{pi}_{vi}.mp4"
            video_to_cat[video] = cats[(pi + vi) % 3]
            dur = int(rnd(40, 600))
            a = int(rnd(0, dur * 0.6)); b = a + 1 + int(rnd(2, dur * 0.4))
            clip = (pi + vi) % 5 != 0
            mr_gt[prompt][video] = {"clip": clip, "bounds": [a, b], "v_duration": dur}
            mode = (pi + 2 * vi) % 6
            if mode == 0:
                pb = [a, b]                                            # exact
            elif mode == 1:
                pb = [b + 3, b + 9]                                    # disjoint
            elif mode == 2:
                pb = [a, a + (b - a) // 2]                             # IoU near 0.5
            else:
                pb = [max(0, a + int(rnd(-15, 15))), b + int(rnd(-15, 15))]
                if pb[1] <= pb[0]:
                    pb[1] = pb[0] + 1
            mr_pred[prompt][video] = {"bounds": pb}
            if clip:
                n_steps = 2 + int(rnd(0, 7))
                cuts = sorted(set([a, b] + [int(rnd(a + 1, b - 1)) for _ in range(n_steps)]))
                refs = [[cuts[i], cuts[i + 1]] for i in range(len(cuts) - 1)]
                sb_gt[video] = {"bounds": refs}
                preds = []
                for r in refs:                                         # jittered, duplicated, nested and outside boxes
                    if next(it) < 0.8:
                        preds.append([r[0] + int(rnd(-3, 4)), r[1] + int(rnd(-3, 4))])
                    if next(it) < 0.3:
                        preds.append([r[0] + 1, r[1] - 1] if r[1] - r[0] > 3 else [r[0], r[1]])
                if next(it) < 0.5:
                    preds.append([a - 5, a + 2])
                if next(it) < 0.3:
                    preds.append([rnd(a, b), rnd(a, b) + 2.5])         # float bounds
                preds = [p if p[1] > p[0] else [p[0], p[0] + 1] for p in preds]
                sb_pred[video] = {"bounds": preds}
    return {"prompt_to_cat": prompt_to_cat, "video_to_cat": video_to_cat, "mr_gt": mr_gt, "mr_pred": mr_pred,
            "sb_gt": sb_gt, "sb_pred": sb_pred}


def timeline_cases():
    """Inputs of the frame-index <-> timestamp fixture (pure numpy; shared with tests/test_timeline.py): per case the
    frame indices 0..n-1 and a timestamp list holding a regular sweep past the end, every bin value exactly, and the
    doubles just below / above every bin (the digitize(right=True) edge)."""
    cases = []
    for dur in (1.0, 1.9, 2.0, 3.7, 17.3, 59.9, 60.0, 199.99, 200.0, 367.8, 571.4, 1855.2, 2500.5):
        for n in (-1, 1, 2, 3, 20, 32, 64, 300, 2048):
            nn = int(dur) if n < 0 else n
            bins = np.linspace(0, int(dur) - 1, nn)
            t = np.concatenate([np.arange(-1.0, dur + 3.0, 0.37), bins, np.nextafter(bins, -np.inf), np.nextafter(bins, np.inf),
                                np.arange(0, int(dur) + 2, dtype=np.float64)])
            cases.append({"duration": dur, "n_frames": n, "frames": np.arange(nn, dtype=np.int64), "timestamps": t})
    return cases




# ----------------------------------------------------------------------------------
# BASELINE configs[2] sub-corpus of the matched-R@k check: tests/golden/make_golden.py encodes it with the real
# reference; bench.py and the GPU tests re-encode it with the kernels.
# ----------------------------------------------------------------------------------

def c3_corpus(V: int, F: int) -> torch.Tensor:
    """video v's frames = base_v + 0.1 * noise_{v,f} (non-degenerate, seeded): [V, F, 3, 224, 224] fp32."""
    base = frames("c3.base", (V, 1, 3, 224, 224), 5)
    return base + 0.1 * frames("c3.noise", (V, F, 3, 224, 224), 6)


def c3_names(V: int) -> list:
    """File names deliberately NOT in index order under string sort, so the (score, name) tie rule of evaluate.py:58-60 is
    exercised; unique for V <= 257."""
    assert V <= 257
    return [f"vid_{(v * 37) % 257:03d}.mp4" for v in range(V)]


# ----------------------------------------------------------------------------------
# BASELINE configs[2] through the feature-file branch (inference_video_retrieval.py:290-355): one [T_v, E] "feature file" per
# video of a corpus, T_v between 3 and 42 rows (below and above any n_model_frames the tests use, so the linspace
# subsample both repeats and drops rows).  tests/golden/make_golden.py gen_retrieval_run writes these files and runs the
# REAL script over them; the tests regenerate the same tensors.
# ----------------------------------------------------------------------------------

def retrieval_feature_lengths(n_videos: int, seed: int = 9) -> np.ndarray:
    u = uniform_pm1("vr.len", n_videos, seed)
    return (3 + np.floor((u + 1.0) * 20.0)).astype(np.int64).clip(3, 42)


def retrieval_feature_corpus(n_videos: int, embed_dim: int, seed: int = 9) -> list:
    """[n_videos] fp32 tensors [T_v, embed_dim]; rows ~ unit-variance noise around a per-video offset (so pooled rows differ)."""
    lens = retrieval_feature_lengths(n_videos, seed)
    off = np.concatenate([[0], np.cumsum(lens)])
    rows = tensor("vr.rows", (int(off[-1]), embed_dim), 1.0, seed)
    base = tensor("vr.base", (n_videos, embed_dim), 1.0, seed + 1)
    return [rows[off[v]:off[v + 1]] * 0.25 + base[v] for v in range(n_videos)]


def c3_device_block(lo: int, hi: int, n_frames: int, device, dtype=torch.bfloat16) -> torch.Tensor:
    """Videos [lo, hi) of the full-size C3 corpus (base_v + 0.1 * noise_f), generated ON the device from a
    per-video seed — 4096 x 32 frames would be 79 GB of host fp32 — so any partition of the corpus (one sweep, 8 rank
    blocks, 8 real ranks) regenerates identical inputs.  torch's Philox stream: stable for a given torch build, which is
    why the committed digests are keyed by torch.__version__."""
    out = torch.empty((hi - lo, n_frames, 3, 224, 224), device=device, dtype=dtype)
    for i, v in enumerate(range(lo, hi)):
        gen = torch.Generator(device=device)
        gen.manual_seed(100000 + v)
        base = torch.randn((1, 3, 224, 224), device=device, generator=gen)
        out[i] = (base + 0.1 * torch.randn((n_frames, 3, 224, 224), device=device, generator=gen)).to(dtype)
    return out


def c3_device_names(V: int) -> list:
    """Unique file names whose string order is not the corpus order (the tie rule of evaluate.py:58-60 sorts by name)."""
    assert V <= 10007
    return [f"video_{(v * 7919) % 10007:05d}.mp4" for v in range(V)]
