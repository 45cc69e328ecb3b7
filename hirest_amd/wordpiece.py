"""BERT WordPiece tokenizer for the ASR sentence encoder (host side, integer work).

The reference embeds Whisper subtitles with ``SentenceTransformer('sentence-transformers/all-MiniLM-L6-v2')``
(extraction/whisper_ASR/extract_ASR_embedding.py:14,25,54).  Its tokenizer is not under /root/reference: it is
``transformers==4.32.0``'s BERT tokenizer (requirements.txt:5-6) with ``do_lower_case=True`` and the uncased 30 522-entry
vocabulary.  This module restates that published algorithm (normalise: drop control characters, space out CJK, lower-case, strip
combining marks; split on whitespace and punctuation; greedy longest-match-first WordPiece with the ``##`` continuation prefix,
words over 100 characters -> [UNK]) over a caller-supplied ``vocab.txt``; `tests/golden/wordpiece.json` pins it to the
transformers implementation installed in the build container on a synthetic vocabulary (no network: the real vocabulary file
cannot be fetched, the algorithm does not depend on it).
"""
import re
import unicodedata
from typing import Dict, Iterable, List, Sequence

import torch

_SPECIAL = ("[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]")


def _is_whitespace(ch: str) -> bool:
    if ch in (" ", "\t", "\n", "\r"):
        return True
    return unicodedata.category(ch) == "Zs"


def _is_control(ch: str) -> bool:
    if ch in ("\t", "\n", "\r"):
        return False
    return unicodedata.category(ch).startswith("C")


def _is_punctuation(ch: str) -> bool:
    cp = ord(ch)
    if 33 <= cp <= 47 or 58 <= cp <= 64 or 91 <= cp <= 96 or 123 <= cp <= 126:
        return True
    return unicodedata.category(ch).startswith("P")


def _is_cjk(cp: int) -> bool:
    return (0x4E00 <= cp <= 0x9FFF or 0x3400 <= cp <= 0x4DBF or 0x20000 <= cp <= 0x2A6DF or 0x2A700 <= cp <= 0x2B73F
            or 0x2B740 <= cp <= 0x2B81F or 0x2B820 <= cp <= 0x2CEAF or 0xF900 <= cp <= 0xFAFF or 0x2F800 <= cp <= 0x2FA1F)


class WordPieceTokenizer:
    """``tokenizer(sentences)``-style encoder: ``encode_batch`` returns ragged id lists with [CLS] / [SEP] added and the
    sentence-transformers truncation (``max_seq_length`` tokens including the two specials)."""

    def __init__(self, vocab: Iterable[str], do_lower_case: bool = True, max_input_chars_per_word: int = 100):
        self.vocab: Dict[str, int] = {}
        for i, tok in enumerate(vocab):
            tok = tok.rstrip("\n")
            self.vocab.setdefault(tok, i)
        for s in ("[UNK]", "[CLS]", "[SEP]", "[PAD]"):
            if s not in self.vocab:
                raise ValueError(f"vocabulary lacks {s}")
        self.ids_to_tokens = {i: t for t, i in self.vocab.items()}
        self.do_lower_case = do_lower_case
        self.max_chars = max_input_chars_per_word
        self.unk_id, self.cls_id, self.sep_id, self.pad_id = (self.vocab[s] for s in ("[UNK]", "[CLS]", "[SEP]", "[PAD]"))
        self._special_re = re.compile("(" + "|".join(re.escape(s) for s in _SPECIAL if s in self.vocab) + ")")

    @classmethod
    def from_file(cls, vocab_file: str, **kw) -> "WordPieceTokenizer":
        with open(vocab_file, encoding="utf-8") as f:
            return cls(f.readlines(), **kw)

    # ---- normaliser + pre-tokeniser ------------------------------------------------------------------------------
    def _normalize(self, text: str) -> str:
        out = []
        for ch in text:
            cp = ord(ch)
            if cp == 0 or cp == 0xFFFD or _is_control(ch):
                continue
            if _is_whitespace(ch):
                out.append(" ")
            elif _is_cjk(cp):
                out.append(" " + ch + " ")
            else:
                out.append(ch)
        text = "".join(out)
        if self.do_lower_case:
            text = unicodedata.normalize("NFD", text)
            text = "".join(ch for ch in text if unicodedata.category(ch) != "Mn")
            text = text.lower()
        return text

    @staticmethod
    def _split_punct(word: str) -> List[str]:
        pieces, cur = [], []
        for ch in word:
            if _is_punctuation(ch):
                if cur:
                    pieces.append("".join(cur)); cur = []
                pieces.append(ch)
            else:
                cur.append(ch)
        if cur:
            pieces.append("".join(cur))
        return pieces

    def basic_tokens(self, text: str) -> List[str]:
        words = []
        for w in self._normalize(text).split():
            words.extend(self._split_punct(w))
        return words

    # ---- WordPiece ------------------------------------------------------------------------------------------------
    def _wordpiece(self, word: str) -> List[int]:
        if len(word) > self.max_chars:
            return [self.unk_id]
        ids, start, n = [], 0, len(word)
        while start < n:
            end, found = n, None
            while start < end:
                sub = word[start:end] if start == 0 else "##" + word[start:end]
                if sub in self.vocab:
                    found = self.vocab[sub]
                    break
                end -= 1
            if found is None:
                return [self.unk_id]
            ids.append(found)
            start = end
        return ids

    def tokenize_ids(self, text: str) -> List[int]:
        """ids without [CLS] / [SEP].  A special token spelled out in the raw text ("[SEP]") is taken as that token, before any
        normalisation, as the library's added-token matcher does."""
        ids: List[int] = []
        for i, part in enumerate(self._special_re.split(text)):
            if i % 2:
                ids.append(self.vocab[part])
                continue
            for w in self.basic_tokens(part):
                ids.extend(self._wordpiece(w))
        return ids

    def encode(self, text: str, max_seq_length: int = 256) -> List[int]:
        ids = self.tokenize_ids(text)[: max(max_seq_length - 2, 0)]
        return [self.cls_id] + ids + [self.sep_id]

    def encode_batch(self, texts: Sequence[str], max_seq_length: int = 256) -> List[List[int]]:
        return [self.encode(t, max_seq_length) for t in texts]

    def padded(self, texts: Sequence[str], max_seq_length: int = 256):
        """(input_ids, attention_mask) int64 [B, longest], as ``tokenizer(texts, padding=True, truncation='longest_first')``"""
        rows = self.encode_batch(texts, max_seq_length)
        L = max((len(r) for r in rows), default=0)
        ids = torch.full((len(rows), L), self.pad_id, dtype=torch.int64)
        mask = torch.zeros((len(rows), L), dtype=torch.int64)
        for i, r in enumerate(rows):
            ids[i, :len(r)] = torch.tensor(r, dtype=torch.int64)
            mask[i, :len(r)] = 1
        return ids, mask

    def convert_ids_to_tokens(self, ids: Iterable[int]) -> List[str]:
        return [self.ids_to_tokens.get(int(i), "[UNK]") for i in ids]
