"""ASR sentence encoder: the third encoder of the pipeline, on MI355X.

The reference embeds every Whisper subtitle of a video with ``SentenceTransformer('sentence-transformers/all-MiniLM-L6-v2')``
and saves the ``[n_subtitles, 384]`` matrix that ``hirest_dataset.py`` later spreads over the frames
(extraction/whisper_ASR/extract_ASR_embedding.py:14,25,31-34,54,58-59)::

    model = SentenceTransformer(args.model).eval().to('cuda')
    sub_embeddings = model.encode(all_subs, convert_to_tensor=True)

`SentenceTransformer` here keeps that surface (constructor from a model directory, ``.eval()``, ``.to(device)``, ``.encode``) for
the one architecture the reference uses: BERT encoder (6 layers, 384 wide, 12 heads of 32 for MiniLM-L6) -> mean over the
sentence's tokens -> L2 normalise, i.e. sentence-transformers==2.3.0's Transformer / Pooling(mean) / Normalize modules
(requirements.txt:6; not under /root/reference, restated from the published model definition and pinned against
``transformers.BertModel``: tests/golden/minilm_*.npz).

MI355X side: exact fp32 on the joint model's kernels (`hirest_gemm_f32` = v_mfma_f32_32x32x2_f32 with fused bias / GELU / residual
epilogues, `hirest_layernorm`, and the ragged forms `hirest_attention_f32_varlen`, `hirest_embedding_pos_fwd_f32`,
`hirest_pool_l2norm_varlen`).  All sentences of a call are packed row after row into one [tokens, hidden] activation: every GEMM and
LayerNorm is a single launch over all tokens, attention and pooling follow a prefix-offset table — no padding, no attention mask and
no wasted rows exist.  The library pads a batch to its longest sentence and masks the pad keys with an additive -inf-like constant,
which gives pad keys probability exactly 0 — the same numbers.  Heads
narrower than the attention kernel's 64 lanes are zero-padded inside the fused QKV / output weights (exact: the pad lanes add 0).
There is no CPU path: `encode` raises off-GPU.
"""
from __future__ import annotations

import itertools
import json
import os
from typing import Dict, List, Optional, Sequence, Union

import numpy as np
import torch
import torch.nn as nn

from . import _lib, ops
from .wordpiece import WordPieceTokenizer

_AH_MAX = 96   # widest head of hirest_attention_f32 (any multiple of 4 up to it; other widths are zero-padded to the next multiple of 4)


def _load_weights(model_dir: str) -> Dict[str, torch.Tensor]:
    st = os.path.join(model_dir, "model.safetensors")
    if os.path.exists(st):
        from safetensors.torch import load_file
        return load_file(st)
    pt = os.path.join(model_dir, "pytorch_model.bin")
    if os.path.exists(pt):
        return torch.load(pt, map_location="cpu")
    raise FileNotFoundError(f"no model.safetensors / pytorch_model.bin under {model_dir}")


class SentenceTransformer(nn.Module):
    def __init__(self, model_name_or_path: Optional[str] = None, device: Optional[Union[str, torch.device]] = None, *,
                 config: Optional[dict] = None, state_dict: Optional[Dict[str, torch.Tensor]] = None,
                 vocab: Optional[Sequence[str]] = None, max_seq_length: Optional[int] = None):
        """Either a local sentence-transformers model directory (``config.json``, ``model.safetensors`` or
        ``pytorch_model.bin``, ``vocab.txt``, optional ``sentence_bert_config.json``) — the hub name the reference passes cannot
        be downloaded here, so a name that is not a directory raises — or explicit ``config`` + ``state_dict`` (+ ``vocab``)."""
        super().__init__()
        if config is None:
            if model_name_or_path is None or not os.path.isdir(model_name_or_path):
                raise FileNotFoundError(f"{model_name_or_path!r} is not a local model directory (no network access: download "
                                        "sentence-transformers/all-MiniLM-L6-v2 beforehand and pass its path)")
            with open(os.path.join(model_name_or_path, "config.json")) as f:
                config = json.load(f)
            state_dict = _load_weights(model_name_or_path)
            vocab_file = os.path.join(model_name_or_path, "vocab.txt")
            if vocab is None and os.path.exists(vocab_file):
                with open(vocab_file, encoding="utf-8") as f:
                    vocab = f.readlines()
            sb = os.path.join(model_name_or_path, "sentence_bert_config.json")
            if max_seq_length is None and os.path.exists(sb):
                with open(sb) as f:
                    max_seq_length = json.load(f).get("max_seq_length")
        if config.get("hidden_act", "gelu") != "gelu" or config.get("position_embedding_type", "absolute") != "absolute":
            raise NotImplementedError("only the BERT configuration of all-MiniLM-L6-v2 (erf GELU, absolute positions) is built")
        self.config = dict(config)
        self.hidden = int(config["hidden_size"])
        self.layers = int(config["num_hidden_layers"])
        self.heads = int(config["num_attention_heads"])
        self.eps = float(config.get("layer_norm_eps", 1e-12))
        self.dh = self.hidden // self.heads
        if self.dh > _AH_MAX or self.hidden % self.heads or self.hidden % 4:
            raise NotImplementedError(f"head width {self.dh} > {_AH_MAX}")
        self.ah = (self.dh + 3) // 4 * 4               # head width as the kernels see it (MiniLM: 32, no padding) ...
        while (self.heads * self.ah) % 16:             # ... such that the output projection's reduction length suits the GEMM
            self.ah += 4
        self.max_seq_length = int(max_seq_length or 256)                      # all-MiniLM-L6-v2's sentence_bert_config.json
        self.max_seq_length = min(self.max_seq_length, int(config["max_position_embeddings"]))
        self.tokenizer = WordPieceTokenizer(vocab) if vocab is not None else None
        if state_dict is None:
            raise ValueError("state_dict required with an explicit config")
        sd = {k[5:] if k.startswith("bert.") else k: v for k, v in state_dict.items()}
        self._names = []
        for k, v in sd.items():
            if k.endswith("position_ids") or k.endswith("token_type_ids"):
                continue
            # parameter names cannot hold dots: keep the checkpoint's name with '/' and map back in state_dict consumers
            self.register_parameter(k.replace(".", "/"), nn.Parameter(v.detach().float().clone(), requires_grad=False))
            self._names.append(k)
        self._cache = None
        if device is not None:
            self.to(device)

    # nn.Module plumbing: any move / cast invalidates the fused-weight cache
    def _apply(self, fn, *a, **k):
        self._cache = None
        return super()._apply(fn, *a, **k)

    def _p(self, name: str) -> torch.Tensor:
        return self._parameters[name.replace(".", "/")]

    @property
    def device(self) -> torch.device:
        return self._p("embeddings.word_embeddings.weight").device

    def get_sentence_embedding_dimension(self) -> int:
        return self.hidden

    def _w(self):
        if self._cache is not None:
            return self._cache
        dev = self.device
        if dev.type != "cuda":
            raise RuntimeError("hirest_amd.SentenceTransformer runs on MI355X only (no CPU fallback); move the model to a GPU")
        f = lambda n: self._p(n).detach().float().contiguous()
        H, dh, D = self.heads, self.dh, self.hidden
        c = {"word": f("embeddings.word_embeddings.weight"),
             # token type is always 0 (single sentences): fold its row into the position table once
             "pos": (f("embeddings.position_embeddings.weight") + f("embeddings.token_type_embeddings.weight")[0]).contiguous(),
             "eln_w": f("embeddings.LayerNorm.weight"), "eln_b": f("embeddings.LayerNorm.bias")}

        AH = self.ah

        def pad_rows(w, b):      # [H*dh, D] -> [H*AH, D]: head h's rows at AH h .. AH h + dh, zeros after (AH = dh: a copy)
            wp = torch.zeros((H, AH, D), device=dev); wp[:, :dh] = w.view(H, dh, D)
            bp = torch.zeros((H, AH), device=dev); bp[:, :dh] = b.view(H, dh)
            return wp.view(H * AH, D), bp.view(H * AH)
        for i in range(self.layers):
            p = f"encoder.layer.{i}."
            ws, bs = zip(*(pad_rows(f(p + f"attention.self.{n}.weight"), f(p + f"attention.self.{n}.bias"))
                           for n in ("query", "key", "value")))
            c[f"qkv_w.{i}"] = torch.cat(ws, 0).contiguous()
            c[f"qkv_b.{i}"] = torch.cat(bs, 0).contiguous()
            wo = torch.zeros((D, H, AH), device=dev)
            wo[:, :, :dh] = f(p + "attention.output.dense.weight").view(D, H, dh)
            c[f"o_w.{i}"] = wo.view(D, H * AH).contiguous()
            for n in ("attention.output.dense.bias", "attention.output.LayerNorm.weight", "attention.output.LayerNorm.bias",
                      "intermediate.dense.weight", "intermediate.dense.bias", "output.dense.weight", "output.dense.bias",
                      "output.LayerNorm.weight", "output.LayerNorm.bias"):
                c[p + n] = f(p + n)
        self._cache = c
        return c

    # ------------------------------------------------------------------------------------------------------------
    @staticmethod
    def _gemm(a, w, bias, resid=None, act=0):
        lib = _lib.load()
        M, K = a.shape
        N = w.shape[0]
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
        _lib.check(lib.hirest_gemm_f32(a.data_ptr(), K, w.data_ptr(), w.shape[1], bias.data_ptr() if bias is not None else None,
                                       resid.data_ptr() if resid is not None else None, N, None, 0,
                                       out.data_ptr(), N, M, N, K, act, ops.stream_ptr()), "hirest_gemm_f32")
        return out

    def _ln(self, x, w, b):
        return ops.layernorm(x, w, b, self.eps, torch.empty_like(x))

    def _encode_packed(self, ids: torch.Tensor, pos_ids: torch.Tensor, seq_off: torch.Tensor, n: int, max_len: int) -> torch.Tensor:
        """n ragged sentences packed row after row (ids / pos_ids int32 [tokens], seq_off int32 [n + 1], all on the device)
        -> [n, hidden] unit rows.  Every GEMM / LayerNorm runs once over all tokens; only attention and pooling know sentences."""
        c, lib = self._w(), _lib.load()
        D, H, rows = self.hidden, self.heads, ids.numel()
        x = torch.empty((rows, D), dtype=torch.float32, device=ids.device)
        _lib.check(lib.hirest_embedding_pos_fwd_f32(ids.data_ptr(), pos_ids.data_ptr(), c["word"].data_ptr(), c["pos"].data_ptr(),
                                                    x.data_ptr(), rows, D, ops.stream_ptr()), "hirest_embedding_pos_fwd_f32")
        x = self._ln(x, c["eln_w"], c["eln_b"])
        for i in range(self.layers):
            p = f"encoder.layer.{i}."
            qkv = self._gemm(x, c[f"qkv_w.{i}"], c[f"qkv_b.{i}"])
            ctx = torch.empty((rows, H * self.ah), dtype=torch.float32, device=x.device)
            _lib.check(lib.hirest_attention_f32_varlen(qkv.data_ptr(), ctx.data_ptr(), seq_off.data_ptr(), n, max_len, H, self.ah,
                                                       self.dh ** -0.5, 0.0, ops.stream_ptr()), "hirest_attention_f32_varlen")
            a = self._gemm(ctx, c[f"o_w.{i}"], c[p + "attention.output.dense.bias"], resid=x)
            a = self._ln(a, c[p + "attention.output.LayerNorm.weight"], c[p + "attention.output.LayerNorm.bias"])
            h = self._gemm(a, c[p + "intermediate.dense.weight"], c[p + "intermediate.dense.bias"], act=1)
            y = self._gemm(h, c[p + "output.dense.weight"], c[p + "output.dense.bias"], resid=a)
            x = self._ln(y, c[p + "output.LayerNorm.weight"], c[p + "output.LayerNorm.bias"])
        out = torch.empty((n, D), dtype=torch.float32, device=x.device)   # mean over a sentence's tokens, then L2: Pooling + Normalize
        _lib.check(lib.hirest_pool_l2norm_varlen(x.data_ptr(), seq_off.data_ptr(), out.data_ptr(), n, D, ops.stream_ptr()),
                   "hirest_pool_l2norm_varlen")
        return out

    @torch.no_grad()
    def encode_ids(self, rows: Sequence[Sequence[int]], max_tokens_per_pass: int = 1 << 18, pipeline_tokens: int = 12288) -> torch.Tensor:
        """Ragged token-id rows ([CLS] ... [SEP] each, already truncated) -> [N, hidden] fp32 on the model's device."""
        dev = self.device
        if dev.type != "cuda":
            raise RuntimeError("hirest_amd.SentenceTransformer runs on MI355X only (no CPU fallback); move the model to a GPU")
        maxpos, vocab = self.config["max_position_embeddings"], self.config["vocab_size"]
        out = torch.empty((len(rows), self.hidden), dtype=torch.float32, device=dev)
        if not len(rows):
            return out
        # Host side of a pass (flatten the ragged rows, positions, offsets) in numpy, packed into ONE pinned int32 buffer and uploaded
        # asynchronously, and the batch cut into passes of ~pipeline_tokens so that the host prepares pass k + 1 while the GPU runs
        # pass k (as one pass, 45 k tokens cost 3 ms of Python list work with the GPU idle: 18 % of the call).  Rows are independent
        # and every kernel is batch-invariant, so the cut does not change a bit of the result.
        lens_all = np.fromiter(map(len, rows), dtype=np.int64, count=len(rows))
        bad = np.nonzero((lens_all < 1) | (lens_all > maxpos))[0]
        if bad.size:
            raise ValueError(f"sentence {int(bad[0])}: {int(lens_all[bad[0]])} tokens (1 .. {maxpos})")
        per_pass = min(int(max_tokens_per_pass), max(int(pipeline_tokens), maxpos))
        with torch.cuda.device(dev):
            s = 0
            while s < len(rows):
                cum = np.cumsum(lens_all[s:])
                e = s + max(1, int(np.searchsorted(cum, per_pass, side="right")))
                lens = lens_all[s:e]
                tokens, n = int(cum[e - s - 1]), e - s
                host = torch.empty((2 * tokens + n + 1,), dtype=torch.int32).pin_memory()     # ids | positions | offsets
                buf = host.numpy()
                buf[:tokens] = np.fromiter(itertools.chain.from_iterable(rows[s:e]), dtype=np.int64, count=tokens)
                if int(buf[:tokens].min()) < 0 or int(buf[:tokens].max()) >= vocab:
                    raise ValueError("token id outside the vocabulary")
                off = buf[2 * tokens:]
                off[0] = 0
                off[1:] = np.cumsum(lens)
                buf[tokens:2 * tokens] = np.arange(tokens, dtype=np.int64) - np.repeat(off[:-1].astype(np.int64), lens)
                d = host.to(dev, non_blocking=True)
                out[s:e] = self._encode_packed(d[:tokens], d[tokens:2 * tokens], d[2 * tokens:], n, int(lens.max()))
                s = e
        return out

    def tokenize(self, sentences: Sequence[str]) -> List[List[int]]:
        if self.tokenizer is None:
            raise RuntimeError("no vocabulary was given (vocab.txt): use encode_ids with ids tokenised elsewhere")
        return self.tokenizer.encode_batch([str(s).strip() for s in sentences], self.max_seq_length)

    @torch.no_grad()
    def encode(self, sentences: Union[str, Sequence[str]], batch_size: int = 32, show_progress_bar=None,
               output_value: str = "sentence_embedding", convert_to_numpy: bool = True, convert_to_tensor: bool = False,
               device=None, normalize_embeddings: bool = False):
        """``SentenceTransformer.encode`` for the reference's call (list of subtitles, ``convert_to_tensor=True`` -> one
        ``[N, 384]`` tensor on the model's device, rows in input order).  A single string gives a 1-D result; an empty list
        gives an empty tensor.  ``batch_size`` only shaped the library's padding and has no effect on the numbers;
        ``normalize_embeddings`` is moot because the model's own last module already normalises."""
        if output_value != "sentence_embedding":
            raise NotImplementedError("only sentence embeddings are produced")
        if device is not None and torch.device(device) != self.device:
            self.to(device)
        single = isinstance(sentences, str)
        rows = self.tokenize([sentences] if single else list(sentences))
        emb = self.encode_ids(rows) if rows else torch.empty((0, self.hidden), dtype=torch.float32, device=self.device)
        if single:
            emb = emb[0]
        if convert_to_tensor:
            return emb
        return emb.cpu().numpy() if convert_to_numpy else list(emb)

    def checkpoint_state_dict(self) -> Dict[str, torch.Tensor]:
        """the weights under their checkpoint names"""
        return {n: self._p(n).detach() for n in self._names}
