"""CPU ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product path.

A from-scratch, functional fp32 restatement (plain ``torch`` CPU ops on explicit state
dicts) of the reference's frame/text encoding + cross-modal scoring path.  Only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import this file; ``hirest_amd`` never does (it fails loudly when its HIP library is
missing instead of falling back to anything here).

PARITY PIN: every function below is checked in ``tests/test_oracle_golden.py`` against
golden vectors in ``tests/golden/*.npz`` that were produced by importing the real
reference modules from /root/reference in the build container
(``tests/golden/make_golden.py``, which is committed; the reference itself never
travels).  Observed agreement oracle-vs-reference: <= 3e-6 relative (fp32 rounding /
summation order only).

Each function cites the reference file:line it restates.  Paths are relative to
/root/reference.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


# ----------------------------------------------------------------------------------
# primitives
# ----------------------------------------------------------------------------------

def layer_norm(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float) -> torch.Tensor:
    """nn.LayerNorm over the last dim, biased variance (vit_model.py:159,165,285 eps 1e-6;
    eva_model.py:19-25 eps 1e-5)."""
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def gelu_erf(x: torch.Tensor) -> torch.Tensor:
    """nn.GELU() default = exact erf form (vit_model.py:47,53; eva_model.py:289)."""
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def quick_gelu(x: torch.Tensor) -> torch.Tensor:
    """EVA_clip/model.py:175-177."""
    return x * torch.sigmoid(1.702 * x)


# ----------------------------------------------------------------------------------
# EVA-CLIP vision tower (EVA_clip/vit_model.py)
# ----------------------------------------------------------------------------------

def eva_patch_embed(sd: SD, img: torch.Tensor, patch: int) -> torch.Tensor:
    """PatchEmbed.forward (vit_model.py:185-206): Conv2d(k=s=P)+bias, flatten(2).transpose(1,2);
    then cls/pos prologue of forward_features (vit_model.py:326-334)."""
    B, C, H, W = img.shape
    g = H // patch
    w = sd["visual.patch_embed.proj.weight"]
    D = w.shape[0]
    x = img.reshape(B, C, g, patch, g, patch).permute(0, 2, 4, 1, 3, 5).reshape(B, g * g, C * patch * patch)
    x = x @ w.reshape(D, -1).t() + sd["visual.patch_embed.proj.bias"]
    x = torch.cat([sd["visual.cls_token"].expand(B, 1, D), x], dim=1)
    return x + sd["visual.pos_embed"]


def eva_attention(sd: SD, p: str, h: torch.Tensor, heads: int) -> torch.Tensor:
    """Attention.forward (vit_model.py:120-150).  K has no bias; the bias is added
    BEFORE q is scaled (:124-130); no rel-pos bias (use_rel_pos_bias=False, :254)."""
    B, N, D = h.shape
    dh = D // heads
    bias = torch.cat([sd[p + "attn.q_bias"], torch.zeros(D), sd[p + "attn.v_bias"]])
    qkv = h @ sd[p + "attn.qkv.weight"].t() + bias
    qkv = qkv.reshape(B, N, 3, heads, dh).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * dh ** -0.5, qkv[1], qkv[2]
    a = torch.softmax(q @ k.transpose(-2, -1), dim=-1) @ v
    a = a.transpose(1, 2).reshape(B, N, D)
    return a @ sd[p + "attn.proj.weight"].t() + sd[p + "attn.proj.bias"]


def eva_mlp(sd: SD, p: str, h: torch.Tensor) -> torch.Tensor:
    """Mlp.forward (vit_model.py:56-62)."""
    y = gelu_erf(h @ sd[p + "mlp.fc1.weight"].t() + sd[p + "mlp.fc1.bias"])
    return y @ sd[p + "mlp.fc2.weight"].t() + sd[p + "mlp.fc2.bias"]


def eva_block(sd: SD, i: int, x: torch.Tensor, heads: int, eps: float = 1e-6) -> torch.Tensor:
    """Block.forward, gamma_1 is None branch (vit_model.py:175-178); DropPath = identity in eval."""
    p = f"visual.blocks.{i}."
    x = x + eva_attention(sd, p, layer_norm(x, sd[p + "norm1.weight"], sd[p + "norm1.bias"], eps), heads)
    x = x + eva_mlp(sd, p, layer_norm(x, sd[p + "norm2.weight"], sd[p + "norm2.bias"], eps))
    return x


def eva_encode_image(sd: SD, img: torch.Tensor, cfg: dict, n_layers: Optional[int] = None) -> torch.Tensor:
    """EVA_CLIP.encode_image (eva_model.py:317) -> VisionTransformer.forward (vit_model.py:326-351):
    patch-embed, cls/pos, L blocks, LayerNorm(eps 1e-6), take token 0, head Linear."""
    v = cfg["vision_cfg"]
    assert img.shape[-1] == v["image_size"] and img.shape[-2] == v["image_size"]  # vit_model.py:203
    heads = v["width"] // v["head_width"]  # eva_model.py:291
    L = v["layers"] if n_layers is None else n_layers
    x = eva_patch_embed(sd, img.float(), v["patch_size"])
    for i in range(L):
        x = eva_block(sd, i, x, heads)
    x = layer_norm(x[:, 0], sd["visual.norm.weight"], sd["visual.norm.bias"], 1e-6)
    return x @ sd["visual.head.weight"].t() + sd["visual.head.bias"]


# ----------------------------------------------------------------------------------
# CLIP-style text towers (eva_model.py:177-250 and EVA_clip/model.py:343-356)
# ----------------------------------------------------------------------------------

def _mha_block(sd: SD, p: str, x: torch.Tensor, heads: int, mask: Optional[torch.Tensor], act, eps: float = 1e-5):
    """ResidualAttentionBlock (eva_model.py:110-159 / model.py:180-202) around
    nn.MultiheadAttention: in_proj rows [q;k;v], each (head, dh); scores/sqrt(dh) + additive mask."""
    B, L, D = x.shape
    dh = D // heads
    h = layer_norm(x, sd[p + "ln_1.weight"], sd[p + "ln_1.bias"], eps)
    qkv = h @ sd[p + "attn.in_proj_weight"].t() + sd[p + "attn.in_proj_bias"]
    q, k, v = [t.reshape(B, L, heads, dh).transpose(1, 2) for t in qkv.chunk(3, dim=-1)]
    s = (q * dh ** -0.5) @ k.transpose(-2, -1)
    if mask is not None:
        s = s + mask
    a = (torch.softmax(s, dim=-1) @ v).transpose(1, 2).reshape(B, L, D)
    x = x + a @ sd[p + "attn.out_proj.weight"].t() + sd[p + "attn.out_proj.bias"]
    h = layer_norm(x, sd[p + "ln_2.weight"], sd[p + "ln_2.bias"], eps)
    y = act(h @ sd[p + "mlp.c_fc.weight"].t() + sd[p + "mlp.c_fc.bias"])
    return x + y @ sd[p + "mlp.c_proj.weight"].t() + sd[p + "mlp.c_proj.bias"]


def causal_mask(L: int) -> torch.Tensor:
    """build_attention_mask (eva_model.py:224-230): -inf strictly above the diagonal."""
    return torch.full((L, L), float("-inf")).triu_(1)


def eva_encode_text(sd: SD, tok: torch.Tensor, cfg: dict, n_layers: Optional[int] = None) -> torch.Tensor:
    """EVA_CLIP.encode_text (eva_model.py:320) -> TextTransformer.forward (:232-250)."""
    t = cfg["text_cfg"]
    L = t["layers"] if n_layers is None else n_layers
    x = sd["text.token_embedding.weight"][tok] + sd["text.positional_embedding"]
    mask = causal_mask(tok.shape[1])
    for i in range(L):
        x = _mha_block(sd, f"text.transformer.resblocks.{i}.", x, t["heads"], mask, gelu_erf)
    x = layer_norm(x, sd["text.ln_final.weight"], sd["text.ln_final.bias"], 1e-5)
    x = x[torch.arange(x.shape[0]), tok.argmax(dim=-1)]  # EOT = largest id (:243)
    return x @ sd["text.text_projection"]


def eva_forward(sd: SD, img, tok, cfg: dict):
    """EVA_CLIP.forward (eva_model.py:323-334)."""
    if img is None:
        return eva_encode_text(sd, tok, cfg)
    if tok is None:
        return eva_encode_image(sd, img, cfg)
    return (F.normalize(eva_encode_image(sd, img, cfg), dim=-1),
            F.normalize(eva_encode_text(sd, tok, cfg), dim=-1),
            sd["text.logit_scale"].exp())


# ----------------------------------------------------------------------------------
# OpenAI CLIP ViT as vendored by the reference (EVA_clip/model.py) — BASELINE config 1
# ----------------------------------------------------------------------------------

def openai_encode_image(sd: SD, img: torch.Tensor, c: dict, pip_head: bool = False) -> torch.Tensor:
    """VisionTransformer.forward (model.py:254-273).  NOTE the vendored copy drops the CLS
    token and returns ln_post(patch tokens) @ proj -> [B, grid^2, E] (SURVEY hazard H4)."""
    P, W = c["vision_patch_size"], c["vision_width"]
    B, C, H, _ = img.shape
    g = H // P
    x = img.float().reshape(B, C, g, P, g, P).permute(0, 2, 4, 1, 3, 5).reshape(B, g * g, C * P * P)
    x = x @ sd["visual.conv1.weight"].reshape(W, -1).t()  # conv1 has no bias (model.py:220)
    x = torch.cat([sd["visual.class_embedding"].expand(B, 1, W), x], dim=1) + sd["visual.positional_embedding"]
    x = layer_norm(x, sd["visual.ln_pre.weight"], sd["visual.ln_pre.bias"], 1e-5)
    heads = W // 64  # model.py:299
    for i in range(c["vision_layers"]):
        x = _mha_block(sd, f"visual.transformer.resblocks.{i}.", x, heads, None, quick_gelu)
    if pip_head:
        # NOT the reference tree: the pip `clip` package (openai/CLIP @ a9b1bf5920416aaeaec965c25dd9e8f98c864f16, pinned by
        # requirements.txt:25) ends VisionTransformer.forward with x = self.ln_post(x[:, 0, :]); x = x @ self.proj.  Parity unpinned:
        # no golden vector of that package exists here; this line restates its published forward.
        return layer_norm(x[:, 0], sd["visual.ln_post.weight"], sd["visual.ln_post.bias"], 1e-5) @ sd["visual.proj"]
    x = layer_norm(x[:, 1:], sd["visual.ln_post.weight"], sd["visual.ln_post.bias"], 1e-5)
    return x @ sd["visual.proj"]


def openai_encode_text(sd: SD, tok: torch.Tensor, c: dict) -> torch.Tensor:
    """CLIP.encode_text (model.py:343-356)."""
    x = sd["token_embedding.weight"][tok] + sd["positional_embedding"]
    mask = causal_mask(tok.shape[1])
    for i in range(c["transformer_layers"]):
        x = _mha_block(sd, f"transformer.resblocks.{i}.", x, c["transformer_heads"], mask, quick_gelu)
    x = layer_norm(x, sd["ln_final.weight"], sd["ln_final.bias"], 1e-5)
    return x[torch.arange(x.shape[0]), tok.argmax(dim=-1)] @ sd["text_projection"]


# ----------------------------------------------------------------------------------
# pooling, scoring, ranking (inference_video_retrieval.py:283-334, evaluate.py:33-81)
# ----------------------------------------------------------------------------------

def subsample_ids(n_frames: int, n_model_frames: int) -> np.ndarray:
    """np.linspace(0, n-1, F).astype(int) (inference_video_retrieval.py:39,315)."""
    return np.linspace(0, n_frames - 1, n_model_frames).astype(int)


def dataset_fit_frames(video_features: torch.Tensor, n_model_frames: int) -> torch.Tensor:
    """hirest_dataset.py:333-356, loop for loop: linspace subsample when longer, bucket up-sample otherwise."""
    if n_model_frames <= 0:
        return video_features
    n_frames = video_features.shape[0]
    if n_frames > n_model_frames:
        ids = torch.from_numpy(np.linspace(0, n_frames - 1, n_model_frames).astype(int))
        return video_features[ids]
    x = torch.zeros((n_model_frames, video_features.shape[1]))
    slots = [0] * n_model_frames
    buckets = [slots[(j * n_model_frames) // n_frames:((j + 1) * n_model_frames) // n_frames] for j in range(n_frames)]
    j = 0
    for k in range(len(buckets)):
        for _ in buckets[k]:
            x[j] = video_features[k]
            j += 1
    return x


def dataset_asr_feats(asr_features: torch.Tensor, sub_spans, fitted_len: int, n_model_frames: int) -> torch.Tensor:
    """hirest_dataset.py:358-402: per-second warping on the FITTED video length, then the same frame-count rule."""
    warped = torch.zeros(fitted_len, asr_features.shape[1]).float()
    for i, (start, end) in enumerate(sub_spans):
        warped[start:end] = asr_features[i]
    return dataset_fit_frames(warped, n_model_frames)


def pool_video(frame_embeds: torch.Tensor, normalize_frames_first: bool = False) -> torch.Tensor:
    """[V,F,E] -> [V,E]: mean over frames then L2 (inference_video_retrieval.py:283-285, 323-327).
    ``normalize_frames_first`` reproduces extract_features.py:64 features (hazard H2)."""
    x = frame_embeds.float()
    if normalize_frames_first:
        x = x / x.norm(dim=-1, keepdim=True)
    x = x.mean(dim=1)
    return x / x.norm(dim=-1, keepdim=True)


def l2_normalize(x: torch.Tensor) -> torch.Tensor:
    """text_embeds /= text_embeds.norm(dim=-1, keepdim=True) (inference_video_retrieval.py:212)."""
    x = x.float()
    return x / x.norm(dim=-1, keepdim=True)


def similarity(text_n: torch.Tensor, video_n: torch.Tensor) -> torch.Tensor:
    """torch.matmul(T, V.T) (inference_video_retrieval.py:334)."""
    return text_n @ video_n.t()


def retrieval_run(text_embeds: torch.Tensor, features: Sequence[torch.Tensor], n_model_frames: int) -> torch.Tensor:
    """The feature-file branch of the retrieval driver, loop for loop (inference_video_retrieval.py:207-212, 298-334):
    text rows L2-normalised; every video's [T, E] file subsampled with linspace when n_model_frames > 0 (:311-317), cast to
    float (:319), mean over rows with keepdim (:323), L2 (:326); scores = T @ V.T (:334)."""
    t = text_embeds.float()
    t = t / t.norm(dim=-1, keepdim=True)
    rows = []
    for video_embeds in features:
        if n_model_frames > 0:
            n_frames = video_embeds.shape[0]
            frame_ids = torch.from_numpy(np.linspace(0, n_frames - 1, n_model_frames).astype(int))
            video_embeds = video_embeds[frame_ids]
        video_embeds = video_embeds.float().mean(dim=0, keepdim=True)
        rows.append(video_embeds / video_embeds.norm(dim=-1, keepdim=True))
    return torch.matmul(t, torch.cat(rows, dim=0).T)


def retrieval_output(prompts: Sequence[str], video_ids: Sequence[str], scores: torch.Tensor) -> dict:
    """inference_video_retrieval.py:337-346: every prompt lists ALL video ids (corpus order) and its score row."""
    return {p: {"videos": list(video_ids), "scores": scores[i].tolist()} for i, p in enumerate(prompts)}


def rank_videos(scores_row: Sequence[float], names: Sequence[str]) -> List[str]:
    """evaluate.py:58-60: sorted(zip(scores, videos)) reversed -> descending score, ties broken
    by video NAME descending (hazard H5)."""
    pairs = sorted(zip(scores_row, names))
    return [n for _, n in pairs[::-1]]


def topk_with_ties(scores: torch.Tensor, tie_rank: torch.Tensor, k: int) -> torch.Tensor:
    """Index form of ``rank_videos``: order by (score desc, tie_rank desc).  tie_rank[v] is the
    rank of video v's name in ascending name order."""
    s = scores.double().numpy()
    t = tie_rank.numpy()
    out = np.empty((s.shape[0], k), dtype=np.int64)
    for q in range(s.shape[0]):
        order = np.lexsort((-t, -s[q]))  # last key is primary
        out[q] = order[:k]
    return torch.from_numpy(out)


def recall_at_k(scores: torch.Tensor, names: Sequence[str], gt: Sequence[Sequence[str]],
                ks=(1, 5, 10, 50)) -> Dict[str, float]:
    """evaluate_video_retrieval, 'all' category (evaluate.py:33-81)."""
    count = {k: 0 for k in ks}
    rows = scores.tolist()
    for q, row in enumerate(rows):
        ranked = rank_videos(row, names)
        g = set(gt[q])
        for k in ks:
            if any(v in g for v in ranked[:k]):
                count[k] += 1
    return {f"R@{k}": count[k] / len(rows) * 100 for k in ks}


# ----------------------------------------------------------------------------------
# timestamp <-> frame index and IoU (hirest_dataset.py:12-68, evaluate.py:25-31)
# ----------------------------------------------------------------------------------

def frame_index_to_timestamp(frame_index: int, v_duration: float, n_frames: int) -> int:
    """hirest_dataset.py:42-68 (n_frames < 0 means one frame per second)."""
    d = int(v_duration)
    n = d if n_frames < 0 else n_frames
    bins = np.linspace(0, d - 1, n)
    return int(bins[frame_index])


def timestamp_to_frame_index(t: float, v_duration: float, n_frames: int) -> int:
    """hirest_dataset.py:12-40: digitize(right=True), clamped to n-1."""
    d = int(v_duration)
    n = d if n_frames < 0 else n_frames
    bins = np.linspace(0, d - 1, n)
    idx = int(np.digitize(t, bins, right=True))
    return min(idx, n - 1)


def compute_iou(a, b) -> float:
    """evaluate.py:25-31."""
    inter = max(0, min(b[1], a[1]) - max(b[0], a[0]))
    union = min(max(b[1], a[1]) - min(b[0], a[0]), b[1] - b[0] + a[1] - a[0])
    return float(inter) / (union + 1e-8)


# ----------------------------------------------------------------------------------
# Joint model: fusion + 2-layer BERT-style encoder + heads (modeling.py:155-224, module_visual.py)
# ----------------------------------------------------------------------------------

_V = "clip4cap_model.visual."


def joint_time_grid(vis_mask: torch.Tensor) -> torch.Tensor:
    """modeling.py:176-193: per sample (linspace(0,1,n)-0.5)*2 over its n valid frames, zero padded."""
    B, T = vis_mask.shape
    out = torch.zeros(B, T)
    for b in range(B):
        n = int(vis_mask[b].sum())
        out[b, :n] = (torch.linspace(0, 1, n) - 0.5) * 2
    return out


def joint_fusion(sd: SD, vis, text, asr, vis_mask, moment_mask, boundary_mask=None) -> torch.Tensor:
    """foward_moment_shared up to the encoder input (modeling.py:155-199).  TF-style LayerNorm eps 1e-12
    (until_module.py:40-53) on the mapped frames, torch LayerNorm (eps 1e-5) inside asr_enc_layer."""
    v = vis @ sd["clip_g_map.weight"].t() + sd["clip_g_map.bias"]
    v = layer_norm(v, sd["clip4cap_model.normalize_video.visual_norm2d.weight"],
                   sd["clip4cap_model.normalize_video.visual_norm2d.bias"], 1e-12)
    t = text @ sd["clip_g_map_text.weight"].t() + sd["clip_g_map_text.bias"]
    t = t / t.norm(dim=-1, keepdim=True)
    f = v * t.unsqueeze(1)
    a = layer_norm(asr, sd["asr_enc_layer.0.weight"], sd["asr_enc_layer.0.bias"], 1e-5)
    f = f + (a @ sd["asr_enc_layer.1.weight"].t() + sd["asr_enc_layer.1.bias"])
    if boundary_mask is not None:
        f = f + sd["boundary_embed.weight"][boundary_mask]
    tg = joint_time_grid(vis_mask).unsqueeze(-1)
    te = torch.tanh(tg @ sd["temporal_embed.0.weight"].t() + sd["temporal_embed.0.bias"])
    f = f + (te @ sd["temporal_embed.2.weight"].t() + sd["temporal_embed.2.bias"])
    return f + sd["mask_embed.weight"][moment_mask]


def visual_encoder(sd: SD, f: torch.Tensor, heads: int = 12, layers: int = 2) -> torch.Tensor:
    """VisualModel.forward with an all-zeros mask (modeling.py:208 passes zeros, module_visual.py:406-414
    turns that into a UNIFORM -10000 added to every score in fp32 — SURVEY hazard H3 — reproduced here)."""
    B, T, _ = f.shape
    x = f @ sd[_V + "embeddings.word_embeddings.weight"].t() + sd[_V + "embeddings.word_embeddings.bias"]
    x = x + sd[_V + "embeddings.position_embeddings.weight"][:T]
    x = layer_norm(x, sd[_V + "embeddings.LayerNorm.weight"], sd[_V + "embeddings.LayerNorm.bias"], 1e-12)
    D = x.shape[-1]
    dh = D // heads
    for i in range(layers):
        p = _V + f"encoder.layer.{i}."
        q = (x @ sd[p + "attention.self.query.weight"].t() + sd[p + "attention.self.query.bias"]).view(B, T, heads, dh).transpose(1, 2)
        k = (x @ sd[p + "attention.self.key.weight"].t() + sd[p + "attention.self.key.bias"]).view(B, T, heads, dh).transpose(1, 2)
        v = (x @ sd[p + "attention.self.value.weight"].t() + sd[p + "attention.self.value.bias"]).view(B, T, heads, dh).transpose(1, 2)
        s = (q @ k.transpose(-1, -2)) / math.sqrt(dh) + (-10000.0)
        c = (torch.softmax(s, dim=-1) @ v).transpose(1, 2).reshape(B, T, D)
        a = layer_norm(c @ sd[p + "attention.output.dense.weight"].t() + sd[p + "attention.output.dense.bias"] + x,
                       sd[p + "attention.output.LayerNorm.weight"], sd[p + "attention.output.LayerNorm.bias"], 1e-12)
        h = gelu_erf(a @ sd[p + "intermediate.dense.weight"].t() + sd[p + "intermediate.dense.bias"])
        x = layer_norm(h @ sd[p + "output.dense.weight"].t() + sd[p + "output.dense.bias"] + a,
                       sd[p + "output.LayerNorm.weight"], sd[p + "output.LayerNorm.bias"], 1e-12)
    return x


def joint_features(sd: SD, vis, text, asr, vis_mask, moment_mask, boundary_mask=None) -> torch.Tensor:
    return visual_encoder(sd, joint_fusion(sd, vis, text, asr, vis_mask, moment_mask, boundary_mask))


def head_logits(sd: SD, feats: torch.Tensor, name: str) -> torch.Tensor:
    """nn.Linear(768, 1).squeeze(2) (modeling.py:80-99, 218-219, 319)."""
    return (feats @ sd[name + ".0.weight"].t() + sd[name + ".0.bias"]).squeeze(2)


def moment_retrieval(sd: SD, vis, text, asr, vis_mask, moment_mask):
    """test_moment_retrieval (modeling.py:272-310): masked argmax of start / end logits -> [[s, e]]."""
    feats = joint_features(sd, vis, text, asr, vis_mask, moment_mask)
    s, e = head_logits(sd, feats, "start_predictor"), head_logits(sd, feats, "end_predictor")
    s = s.masked_fill(vis_mask == 0, -1e10)
    e = e.masked_fill(vis_mask == 0, -1e10)
    return torch.stack([s.argmax(1), e.argmax(1)], -1).tolist(), s, e


def moment_retrieval_loss(sd: SD, vis, text, asr, vis_mask, moment_mask, start_target, end_target) -> torch.Tensor:
    """train_moment_retrieval (modeling.py:226-270) with dropout off: mean-over-moment-frames BCE of the start / end logits
    against one-hot targets, averaged over the two heads.  Plain differentiable torch: autograd on it is the gradient oracle
    of tests/ (the real reference's gradients are in tests/golden/train_*.npz)."""
    feats = joint_features(sd, vis, text, asr, vis_mask, moment_mask)
    s, e = head_logits(sd, feats, "start_predictor"), head_logits(sd, feats, "end_predictor")
    m = moment_mask.float()
    out = 0.0
    for lg, tgt in ((s, start_target), (e, end_target)):
        onehot = torch.zeros_like(lg).scatter_(1, tgt.unsqueeze(1), 1.0)
        l = torch.nn.functional.binary_cross_entropy_with_logits(lg, onehot, reduction="none") * m
        out = out + l.sum() / m.sum().clamp(min=1)
    return out / 2


def moment_segmentation_loss(sd: SD, vis, text, asr, vis_mask, moment_mask, prev_boundary_mask, target) -> torch.Tensor:
    """train_moment_segmentation (modeling.py:323-351): cross-entropy over frames of the segment logits, frames outside the
    moment filled with -finfo.max (an in-place fill: those frames get no gradient)."""
    feats = joint_features(sd, vis, text, asr, vis_mask, moment_mask, prev_boundary_mask)
    lg = head_logits(sd, feats, "segment_predictor")
    lg = lg.masked_fill(moment_mask == 0, -torch.finfo(lg.dtype).max)
    return torch.nn.functional.cross_entropy(lg, target)


def segmentation_walk(scores: Sequence[float], max_idx: int, threshold: float):
    """The per-sample threshold walk of modeling.py:399-433.  Returns (left, right) or None when skipped."""
    max_score = scores[max_idx]
    if max_score < 0.00001:
        return None
    left = right = max_idx
    while (scores[left] / max_score) > threshold:
        if left == 0:
            break
        left -= 1
    while (scores[right] / max_score) > threshold:
        if right == len(scores) - 1:
            break
        right += 1
    if left == 0 or right == 0:
        return None
    return left, right


def segmentation_postprocess(steps: List[List[int]], last_bound: int) -> List[int]:
    """modeling.py:435-463: sort by start, flatten, drop values past the moment end, set(), sort, then keep
    bounds at least 5 apart — the LAST candidate is never appended (range(1, len-1), modeling.py:458)."""
    steps = sorted(steps, key=lambda x: x[0])
    flat = [v for st in steps for v in st]
    while flat[-1] > last_bound:
        flat.pop(-1)
    temp = sorted(set(flat))
    out = [temp[0]]
    cur = temp[0]
    for i in range(1, len(temp) - 1):
        if temp[i] - cur >= 5:
            out.append(temp[i])
            cur = temp[i]
    return out


def moment_segmentation(sd: SD, vis, text, asr, vis_mask, bounds, threshold: float = 0.5, max_iter: int = 20):
    """test_moment_segmentation (modeling.py:353-474)."""
    B, T = vis_mask.shape
    starts, lasts = bounds[:, 0].tolist(), bounds[:, 1].tolist()
    moment_mask = torch.zeros(B, T, dtype=torch.long)
    boundary_mask = torch.zeros(B, T, dtype=torch.long)
    steps = [[[starts[b], starts[b]]] for b in range(B)]
    for b in range(B):
        moment_mask[b, starts[b]:lasts[b] + 1] = 1
        boundary_mask[b, starts[b]] = 1
    first_logits = None
    for it in range(max_iter):
        feats = joint_features(sd, vis, text, asr, vis_mask, moment_mask, boundary_mask)
        logits = head_logits(sd, feats, "segment_predictor")
        if first_logits is None:
            first_logits = logits.clone()
        logits = logits.masked_fill(moment_mask == 0, -torch.finfo(logits.dtype).max)
        probs = torch.softmax(logits, dim=1)
        amax = probs.argmax(dim=1)
        for b in range(B):
            w = segmentation_walk(probs[b].tolist(), int(amax[b]), threshold)
            if w is None:
                continue
            l, r = w
            moment_mask[b, l:r + 1] = 0
            boundary_mask[b, l] = 1
            boundary_mask[b, r] = 1
            steps[b].append([l, r])
    out = []
    for b in range(B):
        steps[b].append([lasts[b], lasts[b]])
        out.append(segmentation_postprocess(steps[b], lasts[b]))
    return out, first_logits


# ----------------------------------------------------------------------------------
# Step captioning: trim_feats + 2-layer decoder + beam search (modeling.py:529-632,
# clip4caption/modules/module_decoder.py:279-406, modules/beam.py:31-123, clip4caption/train.py:511-599)
# ----------------------------------------------------------------------------------

_DEC = "clip4cap_model.decoder."
BOS_ID, EOS_ID = 101, 102          # '[CLS]' / '[SEP]' of the BERT vocab (beam.py:24-29 via the tokenizer)


def trim_feats(feats: torch.Tensor, moment_mask: torch.Tensor, max_frames: int) -> torch.Tensor:
    """modeling.py:529-554: keep the frames inside the moment; truncate to max_frames, or nearest-neighbour
    upsample with the bucket rule count[(j*F)//N : ((j+1)*F)//N]."""
    out = []
    for b in range(feats.shape[0]):
        z = feats[b][moment_mask[b] == 1]
        N = z.shape[0]
        if max_frames < N:
            z = z[:max_frames]
        else:
            idx = []
            for j in range(N):
                idx += [j] * (((j + 1) * max_frames) // N - (j * max_frames) // N)
            x = torch.zeros((max_frames, z.shape[1]))
            for pos, j in enumerate(idx):
                x[pos] = z[j]
            z = x
        out.append(z)
    return torch.stack(out)


def _mha(sd: SD, p: str, q_in, kv_in, add_mask, heads: int = 12):
    """MultiHeadAttention (module_decoder.py:194-240): scores/sqrt(dh) + additive mask, softmax, context."""
    B, Lq, D = q_in.shape
    Lk = kv_in.shape[1]
    dh = D // heads
    q = (q_in @ sd[p + "att.query.weight"].t() + sd[p + "att.query.bias"]).view(B, Lq, heads, dh).transpose(1, 2)
    k = (kv_in @ sd[p + "att.key.weight"].t() + sd[p + "att.key.bias"]).view(B, Lk, heads, dh).transpose(1, 2)
    v = (kv_in @ sd[p + "att.value.weight"].t() + sd[p + "att.value.bias"]).view(B, Lk, heads, dh).transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) / math.sqrt(dh) + add_mask
    c = (torch.softmax(s, dim=-1) @ v).transpose(1, 2).reshape(B, Lq, D)
    # BertSelfOutput: LayerNorm(dense(ctx) + q_in)  (module_decoder.py:110-123, 262-271)
    return layer_norm(c @ sd[p + "output.dense.weight"].t() + sd[p + "output.dense.bias"] + q_in,
                      sd[p + "output.LayerNorm.weight"], sd[p + "output.LayerNorm.bias"], 1e-12)


def decoder_logits(sd: SD, input_ids: torch.Tensor, enc: torch.Tensor, layers: int = 2, answer_mask=None) -> torch.Tensor:
    """DecoderModel.forward (module_decoder.py:372-406) with answer_mask = ones and encoder_mask = ZEROS
    (modeling.py:591), i.e. a uniform -10000 on every cross-attention score and -10000 above the diagonal of the
    self-attention (not -inf).  Returns [R, t, vocab]."""
    R, t = input_ids.shape
    x = sd[_DEC + "embeddings.word_embeddings.weight"][input_ids] + sd[_DEC + "embeddings.position_embeddings.weight"][:t]
    x = layer_norm(x, sd[_DEC + "embeddings.LayerNorm.weight"], sd[_DEC + "embeddings.LayerNorm.bias"], 1e-12)
    self_mask = torch.triu(torch.ones(t, t), diagonal=1) * -10000.0
    if answer_mask is not None:   # training (module_decoder.py:388-397): padded keys are blocked as well, per sample
        blocked = (torch.triu(torch.ones(t, t), diagonal=1).unsqueeze(0) + (1.0 - answer_mask.float()).unsqueeze(1)).gt(0)
        self_mask = (blocked.float() * -10000.0).unsqueeze(1)               # [R, 1, t, t]
    for i in range(layers):
        p = _DEC + f"decoder.layer.{i}."
        s = _mha(sd, p + "slf_attn.", x, x, self_mask)
        d = _mha(sd, p + "enc_attn.", s, enc, -10000.0)
        h = gelu_erf(d @ sd[p + "intermediate.dense.weight"].t() + sd[p + "intermediate.dense.bias"])
        x = layer_norm(h @ sd[p + "output.dense.weight"].t() + sd[p + "output.dense.bias"] + d,
                       sd[p + "output.LayerNorm.weight"], sd[p + "output.LayerNorm.bias"], 1e-12)
    c = _DEC + "classifier.cls.predictions."
    h = gelu_erf(x @ sd[c + "transform.dense.weight"].t() + sd[c + "transform.dense.bias"])
    h = layer_norm(h, sd[c + "transform.LayerNorm.weight"], sd[c + "transform.LayerNorm.bias"], 1e-12)
    return h @ sd[c + "decoder.weight"].t() + sd[c + "bias"]


def step_captioning_loss(sd: SD, vis, text, asr, moment_mask, input_ids, decoder_mask, output_ids, max_frames: int = 20) -> torch.Tensor:
    """train_step_captioning (modeling.py:476-527) with dropout off: trim_feats, fusion + encoder on max_frames frames with all-ones
    masks, teacher-forced decoder, CrossEntropyLoss(ignore_index=-1) over the vocabulary."""
    B = vis.shape[0]
    v = trim_feats(vis, moment_mask, max_frames)
    a = trim_feats(asr, moment_mask, max_frames)
    ones = torch.ones((B, max_frames), dtype=torch.long)
    enc = joint_features(sd, v, text, a, ones, ones)
    logits = decoder_logits(sd, input_ids, enc, answer_mask=decoder_mask)
    return torch.nn.functional.cross_entropy(logits.reshape(-1, logits.shape[-1]), output_ids.reshape(-1), ignore_index=-1)


def sentence_embedding(sd: SD, ids: Sequence[int], heads: int, eps: float = 1e-12) -> torch.Tensor:
    """One sentence through sentence-transformers/all-MiniLM-L6-v2 as extraction/whisper_ASR/extract_ASR_embedding.py:25,54 runs
    it (``SentenceTransformer(...).encode``).  The model code is not under /root/reference: it is sentence-transformers==2.3.0
    (requirements.txt:6) = transformers' BertModel -> mean pooling over the attention mask -> L2 normalise
    (modules.json of the model: Transformer, Pooling(mean tokens), Normalize).  Restated from the published BERT equations
    (post-LN encoder, erf GELU, scores / sqrt(dh), token type 0, absolute positions); pinned by tests/golden/minilm_*.npz, which
    make_golden.py produced with the installed ``transformers.BertModel`` on padded batches (so the fixture also shows that the
    padding + additive mask of a batch does not change a sentence's row).  ids: [CLS] ... [SEP] of ONE sentence -> [D]."""
    E = "embeddings."
    t = torch.as_tensor(ids, dtype=torch.int64)
    L = t.numel()
    x = sd[E + "word_embeddings.weight"][t] + sd[E + "token_type_embeddings.weight"][0] + sd[E + "position_embeddings.weight"][:L]
    x = layer_norm(x, sd[E + "LayerNorm.weight"], sd[E + "LayerNorm.bias"], eps)
    D = x.shape[1]
    dh = D // heads
    i = 0
    while f"encoder.layer.{i}.attention.self.query.weight" in sd:
        p = f"encoder.layer.{i}."
        lin = lambda n, v: F.linear(v, sd[p + n + ".weight"], sd[p + n + ".bias"])
        q = lin("attention.self.query", x).view(L, heads, dh).transpose(0, 1)
        k = lin("attention.self.key", x).view(L, heads, dh).transpose(0, 1)
        v = lin("attention.self.value", x).view(L, heads, dh).transpose(0, 1)
        pr = torch.softmax(q @ k.transpose(1, 2) / math.sqrt(dh), dim=-1)
        ctx = (pr @ v).transpose(0, 1).reshape(L, D)
        a = layer_norm(lin("attention.output.dense", ctx) + x, sd[p + "attention.output.LayerNorm.weight"],
                       sd[p + "attention.output.LayerNorm.bias"], eps)
        h = gelu_erf(lin("intermediate.dense", a))
        x = layer_norm(lin("output.dense", h) + a, sd[p + "output.LayerNorm.weight"], sd[p + "output.LayerNorm.bias"], eps)
        i += 1
    pooled = x.sum(0) / max(float(L), 1e-9)
    return pooled / pooled.norm().clamp_min(1e-12)


def sentence_embeddings(sd: SD, rows: Sequence[Sequence[int]], heads: int) -> torch.Tensor:
    D = sd["embeddings.LayerNorm.weight"].numel()
    if not len(rows):
        return torch.zeros((0, D))
    return torch.stack([sentence_embedding(sd, r, heads) for r in rows])


class RefBeam:
    """beam.py:31-123 restated on plain Python lists (scores are fp32 values)."""

    def __init__(self, size: int):
        self.size, self.done = size, False
        self.scores = torch.zeros(size)
        self.prev_ks: List[List[int]] = []
        self.next_ys: List[List[int]] = [[BOS_ID] * size]

    def hypothesis(self, k: int) -> List[int]:
        hyp = []
        for j in range(len(self.prev_ks) - 1, -1, -1):
            hyp.append(self.next_ys[j + 1][k])
            k = self.prev_ks[j][k]
        return hyp[::-1]

    def current_state(self) -> List[List[int]]:
        if len(self.next_ys) == 1:
            return [[BOS_ID] for _ in range(self.size)]
        keys = torch.sort(self.scores, 0, True)[1].tolist()
        return [[BOS_ID] + self.hypothesis(k) for k in keys]

    def advance(self, word_logprob: torch.Tensor) -> bool:
        V = word_logprob.shape[1]
        lk = word_logprob + self.scores.unsqueeze(1) if self.prev_ks else word_logprob[0]
        best, ids = lk.reshape(-1).topk(self.size, 0, True, True)
        self.scores = best
        prev = (ids // V).tolist()
        self.prev_ks.append(prev)
        self.next_ys.append((ids - (ids // V) * V).tolist())
        if self.next_ys[-1][0] == EOS_ID:
            self.done = True
        return self.done


def step_captioning(sd: SD, vis, text, asr, moment_mask, beams: int = 5, max_frames: int = 20, max_words: int = 48):
    """test_step_captioning (modeling.py:556-632): returns the best hypothesis (token ids) per sample."""
    B = vis.shape[0]
    v = trim_feats(vis, moment_mask, max_frames)
    a = trim_feats(asr, moment_mask, max_frames)
    ones = torch.ones((B, max_frames), dtype=torch.long)
    enc = joint_features(sd, v, text, a, ones, ones)
    bms = [RefBeam(beams) for _ in range(B)]
    active = list(range(B))
    for t in range(1, max_words + 1):
        seqs = torch.tensor([s for b in active for s in bms[b].current_state()], dtype=torch.long)
        enc_rpt = torch.cat([enc[b:b + 1].expand(beams, -1, -1) for b in active], 0)
        logp = torch.log_softmax(decoder_logits(sd, seqs, enc_rpt)[:, -1, :], dim=1).view(len(active), beams, -1)
        active = [b for i, b in enumerate(active) if not bms[b].advance(logp[i])]
        if not active:
            break
    out = []
    for b in range(B):
        k = torch.sort(bms[b].scores, 0, True)[1][0].item()
        out.append(bms[b].hypothesis(k))
    return out, enc
