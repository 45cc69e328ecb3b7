"""ASR sentence encoder (SURVEY 8f-4 tail; extraction/whisper_ASR/extract_ASR_embedding.py:25,54): WordPiece tokenizer and the
MiniLM BERT -> mean pool -> L2 pipeline.

CPU: tokenizer and oracle against fixtures produced by the transformers library in the build container
(tests/golden/make_golden.py: gen_wordpiece, gen_minilm).  GPU: hirest_amd.SentenceTransformer against the same fixtures."""
import json
import os
import sys

import numpy as np
import pytest
import torch

from hirest_amd import synth
from hirest_amd.wordpiece import WordPieceTokenizer

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
GOLD = os.path.join(HERE, "golden")
CASES = {"minilm_tiny": synth.MINILM_TINY, "minilm_l6": synth.MINILM_L6}


def _rows(g):
    off = np.concatenate([[0], np.cumsum(g["lens"])])
    return [[int(x) for x in g["ids"][off[i]:off[i + 1]]] for i in range(len(g["lens"]))]


@pytest.fixture(scope="module")
def wp():
    with open(os.path.join(GOLD, "wordpiece.json"), encoding="utf-8") as f:
        return json.load(f)


def test_wordpiece_matches_transformers(wp):
    tk = WordPieceTokenizer(wp["vocab"])
    assert len(wp["texts"]) >= 80
    for t, ids, ids8 in zip(wp["texts"], wp["ids"], wp["ids_max8"]):
        assert tk.encode(t) == ids, t
        assert tk.encode(t, 8) == ids8, t
    # every branch is in the fixture: empty text, accents, CJK, control characters, > 100-character word, [UNK], literal specials
    flat = [i for r in wp["ids"] for i in r]
    assert tk.unk_id in flat and wp["ids"][0] == [tk.cls_id, tk.sep_id]
    ids, mask = tk.padded(wp["texts"][:5])
    assert ids.shape == mask.shape and ids.shape[1] == max(len(r) for r in wp["ids"][:5])
    assert [int(m) for m in mask.sum(1)] == [len(r) for r in wp["ids"][:5]]
    assert tk.convert_ids_to_tokens(wp["ids"][2])[0] == "[CLS]"


def test_wordpiece_vocab_file_and_errors(tmp_path, wp):
    p = tmp_path / "vocab.txt"
    p.write_text("\n".join(wp["vocab"]) + "\n", encoding="utf-8")
    tk = WordPieceTokenizer.from_file(str(p))
    assert tk.encode("Hello, World!") == wp["ids"][wp["texts"].index("Hello, World!")]
    with pytest.raises(ValueError):
        WordPieceTokenizer(["a", "b"])


@pytest.mark.parametrize("name", list(CASES))
def test_oracle_sentence_embeddings_match_transformers(name):
    from oracle import ref_cpu
    cfg = CASES[name]
    g = np.load(os.path.join(GOLD, name + ".npz"))
    rows = _rows(g)
    assert min(len(r) for r in rows) == 2 and (name != "minilm_l6" or max(len(r) for r in rows) == 256)
    sd = synth.bert_state_dict(cfg, int(g["seed"]))
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    got = ref_cpu.sentence_embeddings(sd, rows, cfg["num_attention_heads"]).numpy()
    assert got.shape == g["emb"].shape
    assert np.abs(got - g["emb"]).max() < 2e-6
    assert np.allclose(np.linalg.norm(got, axis=1), 1.0, atol=1e-6)


def test_sentence_transformer_loading_surface(tmp_path, wp):
    from safetensors.torch import save_file
    from hirest_amd.sentence_encoder import SentenceTransformer
    with pytest.raises(FileNotFoundError):
        SentenceTransformer("sentence-transformers/all-MiniLM-L6-v2")      # hub names cannot be fetched
    cfg = dict(synth.MINILM_TINY)
    sd = synth.bert_state_dict(cfg, 51)
    d = tmp_path / "model"
    d.mkdir()
    (d / "config.json").write_text(json.dumps(cfg))
    (d / "sentence_bert_config.json").write_text(json.dumps({"max_seq_length": 32, "do_lower_case": False}))
    (d / "vocab.txt").write_text("\n".join(wp["vocab"]) + "\n", encoding="utf-8")
    save_file({("bert." + k if i % 2 else k): v.contiguous() for i, (k, v) in enumerate(sd.items())}, str(d / "model.safetensors"))
    m = SentenceTransformer(str(d)).eval()
    assert m.max_seq_length == 32 and m.get_sentence_embedding_dimension() == 64
    back = m.checkpoint_state_dict()
    assert set(back) == set(sd) and all(torch.equal(back[k], sd[k]) for k in sd)
    assert not any(p.requires_grad for p in m.parameters())
    rows = m.tokenize(["Hello, World!", "  so we're going to add the eggs "])
    assert rows[0] == wp["ids"][wp["texts"].index("Hello, World!")] and len(rows[1]) <= 32
    with pytest.raises(RuntimeError, match="MI355X only"):
        m.encode(["no CPU path"], convert_to_tensor=True)


# ------------------------------------------------------------------------------------------------------------ GPU
@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(CASES))
def test_gpu_sentence_embeddings_match_transformers(dev, name):
    from hirest_amd.sentence_encoder import SentenceTransformer
    cfg = CASES[name]
    g = np.load(os.path.join(GOLD, name + ".npz"))
    rows = _rows(g)
    m = SentenceTransformer(config=cfg, state_dict=synth.bert_state_dict(cfg, int(g["seed"]))).eval().to(dev)
    got = m.encode_ids(rows)
    assert got.device == dev and got.dtype == torch.float32 and tuple(got.shape) == g["emb"].shape
    err = np.abs(got.cpu().numpy() - g["emb"]).max()
    cos = (got.cpu().numpy() * g["emb"]).sum(1).min()
    print(f"\n  {name}: {len(rows)} sentences, lengths {min(map(len, rows))}..{max(map(len, rows))}: max |diff| {err:.2e}, min cos {cos:.8f}")
    assert err < 5e-6 and cos > 1 - 1e-6
    # input order does not matter (sentences are regrouped by length internally)
    perm = list(reversed(range(len(rows))))
    again = m.encode_ids([rows[i] for i in perm])
    assert torch.equal(again, got[torch.tensor(perm, device=dev)])


@pytest.mark.gpu
def test_gpu_encode_strings_like_the_reference_call(dev, wp):
    """model.encode(all_subs, convert_to_tensor=True) (extract_ASR_embedding.py:54) on text, checked against the oracle on the
    ids the (transformers-pinned) tokenizer produces"""
    from oracle import ref_cpu
    from hirest_amd.sentence_encoder import SentenceTransformer
    cfg = dict(synth.MINILM_TINY)
    sd = synth.bert_state_dict(cfg, 51)
    m = SentenceTransformer(config=cfg, state_dict=sd, vocab=wp["vocab"], max_seq_length=40).eval().to(dev)
    subs = wp["texts"][:40]
    emb = m.encode(subs, convert_to_tensor=True)
    assert emb.device == dev and tuple(emb.shape) == (len(subs), 64)
    ref = ref_cpu.sentence_embeddings(sd, m.tokenize(subs), cfg["num_attention_heads"])
    assert (emb.cpu() - ref).abs().max().item() < 5e-6
    one = m.encode(subs[2], convert_to_tensor=True)
    assert one.dim() == 1 and torch.equal(one, emb[2])
    assert tuple(m.encode([], convert_to_tensor=True).shape) == (0, 64)
    arr = m.encode(subs[:3])
    assert isinstance(arr, np.ndarray) and np.array_equal(arr, emb[:3].cpu().numpy())
    with pytest.raises(ValueError):
        m.encode_ids([[101, 10 ** 6, 102]])
