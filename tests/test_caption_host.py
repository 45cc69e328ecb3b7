"""Host-side pieces of step captioning.  MomentModel._trim_index_table (the vectorised form test_step_captioning uses) against the per-sample list walk _trim_index, which restates
trim_feats (modeling.py:529-554): more selected frames than slots -> the first max_frames; fewer -> frame j repeated
(j + 1) F // N - j F // N times; none -> zero rows (-1)."""
import random

import torch

from hirest_amd.moment_model import MomentModel


def test_trim_index_table_equals_the_list_walk():
    rng = random.Random(7)
    for trial in range(600):
        B, T, F = rng.randint(1, 6), rng.randint(1, 90), rng.choice([1, 2, 5, 20, 33])
        dens = rng.random()
        mask = torch.tensor([[1 if rng.random() < dens else 0 for _ in range(T)] for _ in range(B)], dtype=torch.long)
        if trial % 7 == 0:
            mask[0] = 0                                   # an empty moment
        if trial % 11 == 0:
            mask[-1] = 1                                  # every frame selected
        if trial % 5 == 0:
            mask[0] = 2 * (mask[0] > 0)                   # only the value 1 selects (mask == 1 in the reference)
        want = [MomentModel._trim_index(r, F) for r in mask.tolist()]
        assert MomentModel._trim_index_table(mask, F).tolist() == want, (B, T, F)


def test_trim_index_closed_form_examples():
    # 3 selected frames into 20 slots: repeats 6, 7, 7 (floor differences of j * 20 / 3)
    mask = torch.zeros(1, 10, dtype=torch.long)
    mask[0, [2, 5, 6]] = 1
    assert MomentModel._trim_index_table(mask, 20).tolist()[0] == [2] * 6 + [5] * 7 + [6] * 7
    mask[0, :] = 1
    assert MomentModel._trim_index_table(mask, 4).tolist()[0] == [0, 1, 2, 3]


def test_caption_texts_with_and_without_a_vocabulary():
    """_caption_texts: ids as decimal strings when no BERT vocabulary is attached; with one, the reference's read-out (modeling.py:614-626:
    cut at [SEP] / [PAD], join word pieces)."""
    m = MomentModel.__new__(MomentModel)                      # the method only reads tokenizer_vocab
    m.tokenizer_vocab = None
    assert m._caption_texts([[7, 102, 5], []], True) == {"prediction": ["7 102 5", ""], "token_ids": [[7, 102, 5], []]}
    m.tokenizer_vocab = ["[PAD]", "[CLS]", "[SEP]", "cut", "##ting", "board", "##s"]
    res = m._caption_texts([[3, 4, 5, 6, 2, 3], [5, 0, 3], [2]], False)
    assert res == {"prediction": ["cutting boards", "board", ""]}
