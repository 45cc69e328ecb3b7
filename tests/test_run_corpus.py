"""BASELINE configs[2] as one call: ``hirest_amd.retrieval.run_corpus`` (sources -> shard -> encode / pool -> ONE all-gather ->
score -> the reference's ``{prompt: {"videos", "scores"}}`` dict) and ``hirest_amd.evaluation.evaluate_video_retrieval``.

Pinned by ``tests/golden/retrieval_run.{npz,json}``: what the REAL ``inference_video_retrieval.py`` (its ``__main__`` run by
``make_golden.py retrieval_run`` over the real test split + distractors, 546 prompts x 4282 videos, synthetic feature files,
tiny EVA-CLIP text tower) wrote, and what the REAL ``evaluate.evaluate_video_retrieval`` made of that JSON.

  * CPU (`not gpu`): the oracle restatement reproduces the script's scores / ranks / recall; a world-size-2 **gloo** run of
    ``run_corpus`` on feature files returns, on both ranks, exactly the dict of the 1-rank run (host logic: shard_range,
    FeatureFileSource, RowGather, dict assembly — the arithmetic comes from the oracle through a shim, since the product has
    no CPU path).
  * GPU: the same run on the kernels against the golden; 8 rank blocks of a 512-video x 32-frame g/14 corpus == the single
    sweep bit for bit, and equal to the committed hash of the 1-rank result."""
import hashlib
import json
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from hirest_amd import retrieval, synth

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")


def _golden():
    g = np.load(os.path.join(GOLD, "retrieval_run.npz"))
    j = json.load(open(os.path.join(GOLD, "retrieval_run.json")))
    prompts = json.load(open(os.path.join(GOLD, "test_prompts.json")))
    return g, j, prompts


def _write_feature_files(dirname, video_ids, embed_dim):
    os.makedirs(dirname, exist_ok=True)
    for vid, f in zip(video_ids, synth.retrieval_feature_corpus(len(video_ids), embed_dim)):
        torch.save(f.clone(), os.path.join(dirname, f"{vid}.pt"))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


# ------------------------------------------------------------------------------------------------ oracle vs the real script

def test_oracle_reproduces_the_real_retrieval_script_and_evaluation():
    from hirest_amd.tokenizer import tokenize
    from oracle import eval_cpu as E
    from oracle import ref_cpu as O
    g, j, prompts = _golden()
    cfg = synth.EVA_CLIP_TINY
    sd = synth.eva_clip_state_dict(cfg, int(g["seed"]), towers=("text",))
    te = O.eva_encode_text(sd, tokenize(prompts), cfg)
    feats = synth.retrieval_feature_corpus(len(j["video_ids"]), cfg["embed_dim"])
    scores = O.retrieval_run(te, feats, int(g["n_model_frames"]))
    assert scores.shape == (len(prompts), len(j["video_ids"]))
    assert (scores[:24] - torch.from_numpy(g["scores_head"])).abs().max().item() < 2e-6
    top11 = torch.from_numpy(g["top11_scores"].copy())
    assert (scores.sort(dim=1, descending=True).values[:, :11] - top11).abs().max().item() < 2e-6
    pred = O.retrieval_output(prompts, j["video_ids"], scores)
    assert list(pred) == prompts and pred[prompts[3]]["videos"] == j["video_ids"]
    # ranks: identical wherever the script's own adjacent gaps exceed the fp32 noise between two CPU runs
    gaps = (top11[:, :-1] - top11[:, 1:]).min(dim=1).values
    clear = gaps > 1e-5
    assert clear.float().mean().item() > 0.9
    names = j["video_ids"]
    want = g["top10"]
    for q in torch.nonzero(clear).flatten().tolist():
        assert [names.index(v) for v in O.rank_videos(pred[prompts[q]]["scores"], names)[:10]] == want[q].tolist()
    # evaluate.py:33-81 restated == the real function's result on the script's JSON (the oracle's scores rank the same)
    gt = {p: {v: {} for v in j["gt"][p]} for p in prompts}
    res = E.evaluate_video_retrieval(gt, pred, j["prompt_to_cat"])
    assert res == j["recall"]                                    # every category, every k, to the last bit


# ------------------------------------------------------------------------------------------------ world-size-2 gloo run

class _OracleOps:
    """Arithmetic stand-in for hirest_amd.ops in CPU tests (the product has no CPU path)."""

    @staticmethod
    def pool_l2norm(x, normalize_frames_first=False):
        from oracle import ref_cpu as O
        return O.pool_video(x, normalize_frames_first)

    @staticmethod
    def similarity(t, v):
        from oracle import ref_cpu as O
        return O.similarity(t, v)


class _OracleTextModel:
    embed_dim = synth.EVA_CLIP_TINY["embed_dim"]

    def __init__(self, seed):
        self.sd = synth.eva_clip_state_dict(synth.EVA_CLIP_TINY, seed, towers=("text",))

    def encode_text(self, tok):
        from oracle import ref_cpu as O
        return O.eva_encode_text(self.sd, tok, synth.EVA_CLIP_TINY)


def _corpus_worker(rank, world, port, feat_dir, video_ids, prompts, n_frames, seed, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    real_ops = retrieval.ops
    try:
        retrieval.ops = _OracleOps
        src = retrieval.FeatureFileSource(feat_dir, video_ids, videos_per_call=50)
        res = retrieval.run_corpus(_OracleTextModel(seed), src, prompts, n_frames, device="cpu")
        path = res.save(f"rank{rank}_of_{world}", out_dir)
        assert json.load(open(path)) == dict(res)
    finally:
        retrieval.ops = real_ops        # (the 1-rank reference run is in this process: later GPU tests need the real kernels)
        if world > 1:
            dist.destroy_process_group()


@pytest.mark.parametrize("V", [301, 2])
def test_gloo_world2_run_corpus_equals_one_rank(tmp_path, V):
    g, j, prompts = _golden()
    video_ids, prompts = j["video_ids"][:V], prompts[:40]
    feat_dir, out_dir = str(tmp_path / "feats"), str(tmp_path / "VR_results")
    _write_feature_files(feat_dir, video_ids, synth.EVA_CLIP_TINY["embed_dim"])
    n_frames, seed = int(g["n_model_frames"]), int(g["seed"])
    _corpus_worker(0, 1, 0, feat_dir, video_ids, prompts, n_frames, seed, out_dir)            # the 1-rank run, in process
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_corpus_worker, args=(r, 2, port, feat_dir, video_ids, prompts, n_frames, seed, out_dir))
             for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    one = json.load(open(os.path.join(out_dir, "rank0_of_1.json")))
    for r in range(2):
        two = json.load(open(os.path.join(out_dir, f"rank{r}_of_2.json")))
        assert list(two) == prompts and two == one                                       # exactly: names, order, every score
    assert one[prompts[0]]["videos"] == video_ids and len(one[prompts[0]]["scores"]) == V
    if V == 301:                                                                          # and it is the script's result
        head = torch.tensor([one[p]["scores"] for p in prompts[:24]])
        assert (head - torch.from_numpy(g["scores_head"][:, :V])).abs().max().item() < 2e-6


def test_frame_source_subsamples_like_the_frame_dataset():
    """FrameSource with n_model_frames: np.linspace(0, n-1, F).astype(int) on the frame axis (:36-44), blocks of videos."""
    calls = []

    class M:
        embed_dim = 4

    frames = torch.arange(6 * 10, dtype=torch.float32).reshape(6, 10, 1, 1, 1).expand(6, 10, 3, 2, 2).contiguous()
    orig = retrieval.encode_videos
    try:
        retrieval.encode_videos = lambda model, blk: (calls.append(blk.clone()), blk[:, :, 0, 0, 0].mean(1, keepdim=True).expand(-1, 4))[1]
        src = retrieval.FrameSource([f"v{i}" for i in range(6)], lambda lo, hi: frames[lo:hi], videos_per_call=4, min_frames_per_call=0)
        rows = retrieval.corpus_block_rows(M(), src, 0, 1, n_model_frames=4, device="cpu")
        # default: short videos are grouped until a tower call holds >= 256 frames (here: all 6 videos x 4 frames in one call)
        grouped = retrieval.FrameSource([f"v{i}" for i in range(6)], lambda lo, hi: frames[lo:hi], videos_per_call=4)
        assert grouped._per_call(4, "cpu") == 64 and grouped._per_call(None, "cpu") == 26      # ceil(256 / 4), ceil(256 / 10 decoded frames)
        n_before = len(calls)
        rows_g = retrieval.corpus_block_rows(M(), grouped, 0, 1, n_model_frames=4, device="cpu")
        assert [c.shape[0] for c in calls[n_before:]] == [6] and torch.equal(rows_g, rows)
        calls = calls[:n_before]
    finally:
        retrieval.encode_videos = orig
    assert [c.shape[0] for c in calls] == [4, 2] and all(c.shape[1] == 4 for c in calls)
    ids = np.linspace(0, 9, 4).astype(int)
    assert torch.equal(calls[0][:, :, 0, 0, 0], frames[:4, ids, 0, 0, 0])
    assert rows.shape == (6, 4)


# ------------------------------------------------------------------------------------------------ GPU

@pytest.mark.gpu
def test_gpu_run_corpus_feature_files_vs_the_real_script(tmp_path):
    """run_corpus on the kernels (fp32 text tower, pool + L2, similarity, top-k with the name tie rule) == the JSON the real
    inference_video_retrieval.py wrote, and evaluate_video_retrieval on the device == the real evaluate.py on that JSON."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import hirest_amd
    from hirest_amd import evaluation
    g, j, prompts = _golden()
    dev = torch.device("cuda:0")
    video_ids = j["video_ids"]
    feat_dir = str(tmp_path / "feats")
    _write_feature_files(feat_dir, video_ids, synth.EVA_CLIP_TINY["embed_dim"])
    model, _ = hirest_amd.build_eva_model_and_transforms("EVA_CLIP_tiny_test", pretrained=f"synth:{int(g['seed'])}", precision="fp32")
    model = model.to(dev).eval()
    src = retrieval.FeatureFileSource(feat_dir, video_ids)
    res = retrieval.run_corpus(model, src, prompts, int(g["n_model_frames"]))
    assert list(res) == prompts and res[prompts[7]]["videos"] == video_ids
    scores = res.scores.cpu()
    err = (scores[:24] - torch.from_numpy(g["scores_head"])).abs().max().item()
    top11 = torch.from_numpy(g["top11_scores"].copy())
    err = max(err, (scores.sort(dim=1, descending=True).values[:, :11] - top11).abs().max().item())
    print(f"run_corpus vs the real script: max |score error| {err:.2e}")
    assert err < 2e-5
    assert torch.equal(torch.tensor([res[p]["scores"] for p in prompts[:3]]), scores[:3])     # the dict holds the fp32 values
    _, idx = res.topk(10)
    gaps = (top11[:, :-1] - top11[:, 1:]).min(dim=1).values
    clear = gaps > 4 * err
    assert clear.float().mean().item() > 0.9
    assert torch.equal(idx.cpu().long()[clear], torch.from_numpy(g["top10"].astype(np.int64))[clear])
    gt = {p: {v: {} for v in j["gt"][p]} for p in prompts}
    got = evaluation.evaluate_video_retrieval(gt, res, j["prompt_to_cat"])
    again = evaluation.evaluate_video_retrieval(gt, json.loads(json.dumps(dict(res))), j["prompt_to_cat"])   # the JSON route
    assert got == again
    from oracle import eval_cpu as E
    assert got == E.evaluate_video_retrieval(gt, dict(res), j["prompt_to_cat"])               # same scores -> same numbers
    assert got["all"] == j["recall"]["all"] and set(got) == set(j["recall"])
    # the tie-laden score grid of retrieval_eval.json (real evaluate.py's recall on it)
    d = json.load(open(os.path.join(GOLD, "retrieval_eval.json")))
    u = synth.uniform_pm1("eval.scores", len(d["prompts"]) * len(d["names"]), d["scores_seed"]).reshape(len(d["prompts"]), -1)
    grid = np.round(u * 8).astype(np.float32) / 8.0
    pred = {p: {"videos": d["names"], "scores": grid[i].tolist()} for i, p in enumerate(d["prompts"])}
    got = evaluation.evaluate_video_retrieval({p: {v: {} for v in d["gt"][p]} for p in d["prompts"]}, pred)
    assert got["all"] == d["recall"]


C3_HASHES = os.path.join(GOLD, "c3_rank_blocks.json")


def corpus_digest(rows: torch.Tensor, idx: torch.Tensor) -> dict:
    return retrieval.corpus_digest(rows, idx)


@pytest.mark.gpu
def test_gpu_c3_eight_rank_blocks_equal_the_single_sweep():
    """512 videos x 32 frames through EVA-CLIP-g/14 (bf16 hot path): the corpus encoded as the 8 rank blocks of the 8-GPU run
    (each block by ``corpus_block_rows(rank, 8)``, concatenated in rank order = what the all-gather assembles) gives pooled
    rows, scores and top-10 lists bit-identical to the single sweep, and to the committed digest of the 1-rank result."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import hirest_amd
    dev = torch.device("cuda:0")
    V, F = 512, 32
    model = hirest_amd.EVA_CLIP(**synth.EVA_CLIP_G_14).to(dev).eval()
    model.init_random_(seed=1234)
    prompts = json.load(open(os.path.join(GOLD, "test_prompts.json")))
    ids = synth.c3_device_names(V)
    src = retrieval.FrameSource(ids, lambda lo, hi: synth.c3_device_block(lo, hi, F, dev), videos_per_call=32)
    one = retrieval.run_corpus(model, src, prompts, F)
    blocks = [retrieval.corpus_block_rows(model, src, r, 8, F) for r in range(8)]
    assert [b.shape[0] for b in blocks] == [64] * 8
    rows8 = torch.cat(blocks)
    if not torch.equal(rows8, one.video_rows):                       # say WHICH videos (a call holds 32) and by how much before failing
        bad = (rows8 != one.video_rows).any(dim=1).nonzero().flatten().tolist()
        again = torch.cat([retrieval.corpus_block_rows(model, src, r, 8, F) for r in range(8)])
        pytest.fail(f"rank blocks != single sweep in {len(bad)} videos {bad[:24]}, max |d| {(rows8 - one.video_rows).abs().max().item():.3e}; "
                    f"a third sweep equals the second: {torch.equal(again, rows8)}, the first: {torch.equal(again, one.video_rows)}")
    eight = retrieval.score_corpus(one.text_rows, rows8, ids, prompts)
    assert torch.equal(eight.scores, one.scores) and dict(eight) == dict(one)
    _, i1 = one.topk(10)
    _, i8 = eight.topk(10)
    assert torch.equal(i1, i8)
    digest = corpus_digest(one.video_rows, i1)
    key = f"V{V}_F{F}_torch{torch.__version__}"
    known = json.load(open(C3_HASHES)) if os.path.isfile(C3_HASHES) else {}
    print("c3 digest", key, json.dumps(digest))
    os.makedirs("gpurun_out", exist_ok=True)
    with open(os.path.join("gpurun_out", "c3_rank_blocks_digest.json"), "w") as f:
        json.dump({key: digest}, f)
    assert key in known, f"no committed digest for {key}: copy gpurun_out/c3_rank_blocks_digest.json into {C3_HASHES}"
    assert known[key] == digest


@pytest.mark.gpu
@pytest.mark.parametrize("model", ["tiny", "g14"])      # g14: the production kernels (pq256 GEMMs, v3 attention, LN fold) on both ranks, 64 videos
def test_gpu_two_rank_driver_on_one_gpu_equals_one_rank(tmp_path, model):
    """tools/c3_run.py as TWO PROCESSES (launcher, shard_range blocks, RowGather, replicated text tower + ranking, cross-rank agreement
    check) against the same script as one process: identical SHA-256 of the gathered [V, E] rows and of the top-10 table.  The two ranks
    share the one GPU of a test box and gather through gloo (RCCL refuses two ranks on one device) — the transport is the only part of
    the 8-GPU command (`python tools/c3_run.py --gpus 8`) this run does not exercise."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import subprocess
    import sys
    tool = os.path.join(os.path.dirname(HERE), "tools", "c3_run.py")
    common = [sys.executable, tool, "--model", model, "--videos", "64", "--frames", "4", "--block", "16"]   # 64-frame tower calls on every rank
    reports = {}
    for n in (1, 2):
        out = str(tmp_path / f"c3_{n}.json")
        extra = ["--gpus", "2", "--backend", "gloo", "--share-gpu"] if n == 2 else []
        env = dict(os.environ)
        env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
        r = subprocess.run(common + extra + ["--out", out], capture_output=True, text=True, timeout=900, env=env)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        reports[n] = json.load(open(out))
    one, two = reports[1], reports[2]
    assert two["n_gpus"] == 2 and two["rccl_ranks"] == 2 and two["backend"] == "gloo" and two["all_ranks_hold_the_same_rows_and_ranking"]
    assert (two["pooled_sha256"], two["top10_sha256"]) == (one["pooled_sha256"], one["top10_sha256"])
    assert two["result_dict"] == one["result_dict"] and two["result_dict"]["videos_per_prompt"] == 64


def test_score_corpus_without_prompts_or_videos():
    """No prompts / an empty corpus: the script's loops do not run (inference_video_retrieval.py:337-346) — an empty dict, no kernel call."""
    empty = retrieval.score_corpus(torch.zeros((0, 8)), torch.randn(5, 8), [f"v{i}.mp4" for i in range(5)], [])
    assert dict(empty) == {} and tuple(empty.scores.shape) == (0, 5)
    novid = retrieval.score_corpus(torch.randn(2, 8), torch.zeros((0, 8)), [], ["a", "b"])
    assert dict(novid) == {"a": {"videos": [], "scores": []}, "b": {"videos": [], "scores": []}}
    assert retrieval.encode_prompts(object(), [], "cpu").shape[0] == 0
