"""Host-side beam bookkeeping for step captioning (reference: clip4caption/modules/beam.py:31-123 and the helpers
in clip4caption/train.py:511-599).  Integer control logic only: the decoder forward, log-softmax and the top-k over
(beam x vocabulary) run on the GPU; this class consumes the k winners per step."""
from __future__ import annotations

from typing import List

BOS_ID, EOS_ID = 101, 102     # '[CLS]' and '[SEP]' in the BERT vocabulary the reference decoder uses (beam.py:24-29)


class BeamState:
    def __init__(self, size: int):
        self.size = size
        self.done = False
        self.scores: List[float] = [0.0] * size          # fp32 values as Python floats
        self.backptr: List[List[int]] = []               # prev_ks
        self.tokens: List[List[int]] = [[BOS_ID] * size]  # next_ys

    def _order(self) -> List[int]:
        """torch.sort(scores, descending) order; scores come out of a sorted top-k, so this is the identity
        except for exact ties."""
        return sorted(range(self.size), key=lambda k: -self.scores[k])

    def hypothesis(self, k: int) -> List[int]:
        out = []
        for j in range(len(self.backptr) - 1, -1, -1):
            out.append(self.tokens[j + 1][k])
            k = self.backptr[j][k]
        return out[::-1]

    def current_state(self) -> List[List[int]]:
        """get_tentative_hypothesis (beam.py:100-112): [BOS] + hypothesis for each beam, best first."""
        if len(self.tokens) == 1:
            return [[BOS_ID] for _ in range(self.size)]
        return [[BOS_ID] + self.hypothesis(k) for k in self._order()]

    def advance(self, best_scores: List[float], best_flat_ids: List[int], vocab: int) -> bool:
        """beam.py:70-92 given the device's top-`size` of (beam x vocab): flat id -> (source beam, word)."""
        self.scores = list(best_scores)
        prev = [i // vocab for i in best_flat_ids]
        self.backptr.append(prev)
        self.tokens.append([i - p * vocab for i, p in zip(best_flat_ids, prev)])
        if self.tokens[-1][0] == EOS_ID:
            self.done = True
        return self.done

    def best_hypothesis(self) -> List[int]:
        """collect_hypothesis_and_scores(..., n_best=1) (train.py:590-599)."""
        return self.hypothesis(self._order()[0])
