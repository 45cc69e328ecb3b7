"""Margin-guarded re-rank (``retrieval.run_corpus(rank_exact_k=k)``, round 6): the corpus goes through the fast (bf16) tower, only the
videos whose place in some query's top k is within twice the score error go through the precise one.

CPU: the selection rule is host logic on sorted score windows — property: for ANY perturbation of the scores bounded by eps, replacing
the selected columns by their exact values gives exactly the exact matrix's top-k lists.  GPU: the mixed run's top-k lists equal the
all-bf16x3 run's on a tiny tower and on the g/14 production kernels."""
import numpy as np
import pytest
import torch

from hirest_amd import retrieval


def _window(scores: np.ndarray, W: int):
    idx = np.argsort(-scores, axis=1, kind="stable")[:, :W]
    return np.take_along_axis(scores, idx, 1), idx


def test_ambiguous_columns_on_a_hand_made_window():
    eps = 0.01
    #           rank: 1     2      3      4      5     6
    val = np.array([[0.90, 0.895, 0.80, 0.785, 0.70, 0.10]], dtype=np.float32)
    idx = np.array([[7, 3, 9, 1, 4, 0]])
    # k = 1: threshold 0.88 -> candidates {7, 3}, 0.005 apart: both ambiguous
    assert retrieval.ambiguous_columns(val, idx, 1, eps).tolist() == [3, 7]
    # k = 3: threshold 0.78 -> candidates 7, 3, 9, 1; (7, 3) and (9, 1) are closer than 2 eps
    assert retrieval.ambiguous_columns(val, idx, 3, eps).tolist() == [1, 3, 7, 9]
    # k = 2: threshold 0.875: 9 is far below, (7, 3) ambiguous
    assert retrieval.ambiguous_columns(val, idx, 2, eps).tolist() == [3, 7]
    # well separated top 1: nothing to re-encode
    val2 = np.array([[0.9, 0.5, 0.4]], dtype=np.float32)
    assert retrieval.ambiguous_columns(val2, np.array([[2, 0, 1]]), 1, eps).size == 0
    assert retrieval.window_covers(val2, 1, eps, 100) and not retrieval.window_covers(val[:, :2], 1, eps, 100)
    assert retrieval.window_covers(val[:, :2], 1, eps, 2)                 # the window is the whole corpus
    assert retrieval.ambiguous_columns(np.zeros((0, 4), np.float32), np.zeros((0, 4), np.int64), 1, eps).size == 0


@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("k", [1, 5, 10])
def test_mixed_matrix_has_the_exact_top_k_for_any_bounded_perturbation(seed, k):
    rng = np.random.default_rng(seed)
    Q, V = 40, 300
    exact = (rng.standard_normal((Q, V)) * 0.02).astype(np.float32)
    eps = np.float32([1e-4, 1e-3, 4e-3][seed % 3])
    noise = rng.uniform(-1, 1, (Q, V)).astype(np.float32) * eps
    if seed % 2:                                                            # adversarial: push neighbours towards each other
        order = np.argsort(-exact, axis=1)
        sign = np.where(np.arange(V)[None, :] % 2 == 0, -1.0, 1.0).astype(np.float32)
        np.put_along_axis(noise, order, sign * eps, 1)
    fast = exact + noise
    W = 16
    while True:
        val, idx = _window(fast, W)
        if retrieval.window_covers(val, k, float(eps) * 1.0001, V):
            break
        W = min(V, 2 * W)
    amb = retrieval.ambiguous_columns(val, idx, k, float(eps) * 1.0001)      # (the fp32 sum exact + noise rounds: a hair of slack)
    mixed = fast.copy()
    mixed[:, amb] = exact[:, amb]
    want = np.argsort(-exact, axis=1, kind="stable")[:, :k]
    got = np.argsort(-mixed, axis=1, kind="stable")[:, :k]
    assert np.array_equal(got, want)
    # and the rule is not vacuous: with the larger errors a good part of the corpus stays on the fast rows
    if eps <= 1e-4:
        assert amb.size < V // 2


def test_id_runs_and_rank_split():
    assert retrieval._id_runs([2, 3, 4, 9, 11, 12]) == [(2, 5), (9, 10), (11, 13)]
    ids = list(range(10, 21))
    parts = [retrieval._split_ids(ids, r, 4) for r in range(4)]
    assert sum(parts, []) == ids and [len(p) for p in parts] == [3, 3, 3, 2]
    assert retrieval._split_ids([], 1, 2) == []


def _corpus(model_name, V, F, seed):
    import hirest_amd
    from hirest_amd import synth
    dev = torch.device("cuda:0")
    model, _ = hirest_amd.build_eva_model_and_transforms(model_name, pretrained=f"synth:{seed}", precision="bf16")
    model = model.to(dev).eval()
    frames = synth.c3_corpus(V, F).to(dev)
    names = synth.c3_names(V)
    return model, frames, names, dev


@pytest.mark.gpu
@pytest.mark.parametrize("model_name,V,F,k", [("EVA_CLIP_tiny_test", 96, 4, 5), ("EVA_CLIP_g_14", 64, 4, 3)])
def test_gpu_rank_exact_run_equals_the_all_bf16x3_run(model_name, V, F, k):
    import json, os
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    model, frames, names, dev = _corpus(model_name, V, F, 5)
    prompts = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "test_prompts.json")))[:120]
    src = retrieval.FrameSource(names, frames, videos_per_call=16)
    res = retrieval.run_corpus(model, src, prompts, rank_exact_k=k)
    rep = res.rank_exact
    assert model.visual.precision == "bf16" and 0 < rep["reencoded"] <= V and rep["eps"] > 0
    # the precise run of the whole corpus
    model.set_precision("bf16x3")
    full = retrieval.run_corpus(model, src, prompts)
    _, want = full.topk(k)
    _, got = res.topk(k)
    assert torch.equal(got, want)
    # the bound really bounds: no fast score is further from the precise one than eps
    model.set_precision("bf16")
    fast = retrieval.run_corpus(model, src, prompts)
    model.text.precision = "fp32"
    assert (ops_similarity(res.text_rows, fast.video_rows) - full.scores).abs().max().item() <= rep["eps"]
    print(f"{model_name}: k={k} eps={rep['eps']:.2e} re-encoded {rep['reencoded']} of {V}")


def ops_similarity(t, v):
    from hirest_amd import ops
    return ops.similarity(t.contiguous(), v.contiguous())


@pytest.mark.gpu
def test_gpu_two_rank_rank_exact_run_equals_one_rank(tmp_path):
    """The second pass is sharded too (every rank re-encodes its share of the ambiguous set, one more all-gather): two processes over gloo
    on the one GPU of a test box give the rows, the top-10 table and the re-encoded count of the one-process run."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import json, os, subprocess, sys
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "c3_run.py")
    common = [sys.executable, tool, "--model", "tiny", "--videos", "64", "--frames", "4", "--block", "16", "--rank-exact-k", "3"]
    reports = {}
    for n in (1, 2):
        out = str(tmp_path / f"rx_{n}.json")
        extra = ["--gpus", "2", "--backend", "gloo", "--share-gpu"] if n == 2 else ["--verify-rank-exact"]
        env = dict(os.environ)
        env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
        r = subprocess.run(common + extra + ["--out", out], capture_output=True, text=True, timeout=900, env=env)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        reports[n] = json.load(open(out))
    one, two = reports[1], reports[2]
    assert one["rank_exact"]["topk_lists_equal_all_bf16x3"] is True
    assert two["all_ranks_hold_the_same_rows_and_ranking"] and two["n_gpus"] == 2
    assert (two["pooled_sha256"], two["top10_sha256"]) == (one["pooled_sha256"], one["top10_sha256"])
    assert two["rank_exact"]["reencoded"] == one["rank_exact"]["reencoded"] and two["rank_exact"]["eps"] == one["rank_exact"]["eps"]
