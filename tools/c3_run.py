#!/usr/bin/env python3
"""BASELINE configs[2] at full size, on N ranks: V = 4096 videos x 32 frames (131 072 frames) of the SURVEY 8d corpus
(video v's frames = base_v + 0.1 * noise_f, generated on the device per video), 546 real-prompt queries, EVA-CLIP-g/14
with synthetic weights, through ``hirest_amd.retrieval.run_corpus``: every rank encodes its ``shard_range`` block, ONE
``all_gather_into_tensor`` (RCCL over xGMI) of the pooled rows, scoring + top-10 replicated, rank 0 reports.

    python tools/c3_run.py                      # 1 GPU, 4096 x 32 (about 70 s)
    python tools/c3_run.py --gpus 8             # the 8-GPU run: starts its 8 ranks itself (hirest_amd.launch), one per GPU
    python tools/c3_run.py --videos 512         # the size tests/golden/c3_rank_blocks.json holds a 1-rank digest for

The report (one JSON object on rank 0) holds frames/s, the SHA-256 of the gathered [V, 1024] rows and of the top-10 table,
and — when ``tests/golden/c3_rank_blocks.json`` has a digest for this (V, F, torch build) — ``"equals_committed_1rank_digest"``:
the 1-GPU ranks == N-GPU ranks check of SURVEY 8d, bit for bit.  With ``--rank-blocks R`` (1 rank only) the corpus is also
re-encoded as R rank-sized blocks in this process and compared with the single sweep (the same invariance without an N-GPU
node)."""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from hirest_amd import launch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--videos", type=int, default=4096)
    ap.add_argument("--frames", type=int, default=32)
    ap.add_argument("--block", type=int, default=32, help="videos per encode call (32 x 32 = one 1024-frame tower call)")
    ap.add_argument("--rank-blocks", type=int, default=0, help="1 rank only: also encode the corpus as this many rank blocks")
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--out", default=None, help="also write the report to this path")
    ap.add_argument("--backend", default="nccl", choices=("nccl", "gloo"), help="nccl = RCCL (one rank per GPU); gloo = host-staged gather")
    ap.add_argument("--share-gpu", action="store_true", help="FUNCTIONAL TEST ONLY: all ranks on the GPUs that exist (rank r on device r %% "
                    "device_count; needs --backend gloo: RCCL refuses two ranks on one device).  Exercises the multi-process driver — launcher, "
                    "shard, gather, replicated ranking, digest check — on a one-GPU box; its frames/s is not an N-GPU result and is labelled so")
    ap.add_argument("--model", default="g14", choices=("g14", "tiny"), help="tiny = the 2-layer test config (functional runs)")
    ap.add_argument("--rank-exact-k", type=int, default=0, help="margin-guarded re-rank (retrieval.run_corpus(rank_exact_k=k)): the corpus on the "
                    "--precision tower, the videos whose place in a top k is uncertain again on bf16x3; reports the re-encoded fraction")
    ap.add_argument("--verify-rank-exact", action="store_true", help="also run the whole corpus at bf16x3 and compare the top-k lists (1 rank)")
    a = ap.parse_args()
    import torch
    if a.share_gpu and a.backend != "gloo":
        raise SystemExit("--share-gpu needs --backend gloo (RCCL refuses two ranks on one device)")
    launch.ensure_ranks(a.gpus, os.path.abspath(__file__), sys.argv[1:], visible_devices=None if a.share_gpu else torch.cuda.device_count())
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dev = torch.device("cuda", local_rank % torch.cuda.device_count() if a.share_gpu else local_rank)
    torch.cuda.set_device(dev)
    rank, _, world = launch.init_ranks(a.gpus, a.backend, dev)
    import torch.distributed as dist
    import hirest_amd
    from hirest_amd import retrieval, synth
    model = hirest_amd.EVA_CLIP(**(synth.EVA_CLIP_G_14 if a.model == "g14" else synth.EVA_CLIP_TINY)).to(dev).eval()
    model.init_random_(seed=1234)
    model.set_precision(a.precision)
    prompts = json.load(open(os.path.join(REPO, "tests", "golden", "test_prompts.json")))
    ids = synth.c3_device_names(a.videos)
    src = retrieval.FrameSource(ids, lambda lo, hi: synth.c3_device_block(lo, hi, a.frames, dev), videos_per_call=a.block)
    gather = retrieval.RowGather()

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier(device_ids=[dev.index]) if a.backend == "nccl" else dist.barrier()

    sync()
    t0 = time.perf_counter()
    res = retrieval.run_corpus(model, src, prompts, a.frames, gather=gather, rank_exact_k=a.rank_exact_k)
    val, idx = res.topk(10)
    sync()
    elapsed = time.perf_counter() - t0
    rank_exact = dict(getattr(res, "rank_exact", {}) or {})
    if a.rank_exact_k > 0:
        rank_exact["effective_frames_per_s"] = a.videos * a.frames / elapsed
        if a.verify_rank_exact and world == 1:
            model.set_precision("bf16x3")
            t1 = time.perf_counter()
            full = retrieval.run_corpus(model, src, prompts, a.frames, gather=gather)
            torch.cuda.synchronize(dev)
            rank_exact["all_bf16x3_seconds"] = time.perf_counter() - t1
            rank_exact["all_bf16x3_frames_per_s"] = a.videos * a.frames / rank_exact["all_bf16x3_seconds"]
            kk = min(a.rank_exact_k, a.videos)
            rank_exact["topk_lists_equal_all_bf16x3"] = bool(torch.equal(full.topk(kk)[1], res.topk(kk)[1]))
            rank_exact["top1_flips_vs_all_bf16x3"] = int((full.topk(1)[1] != res.topk(1)[1]).sum())
            model.set_precision(a.precision)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if a.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # every rank must hold the same gathered matrix and the same ranking
        mine = torch.tensor([int(x, 16) % (1 << 62) for x in (retrieval.corpus_digest(res.video_rows, idx)["pooled_sha256"][:15],
                                                               retrieval.corpus_digest(res.video_rows, idx)["top10_sha256"][:15])],
                            dtype=torch.int64, device=dev if a.backend == "nccl" else "cpu")
        all_d = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(all_d, mine)
        ranks_agree = all(torch.equal(d, all_d[0]) for d in all_d)
    else:
        ranks_agree = True
    report = None
    if rank == 0:
        per_rank = -(-a.videos // world)
        whole_calls = a.videos % world == 0 and per_rank % a.block == 0 and a.block * a.frames >= 64
        digest = retrieval.corpus_digest(res.video_rows, idx)
        key = f"V{a.videos}_F{a.frames}_torch{torch.__version__}" + ("" if a.model == "g14" else f"_{a.model}")
        known_path = os.path.join(REPO, "tests", "golden", "c3_rank_blocks.json")
        known = json.load(open(known_path)) if os.path.isfile(known_path) else {}
        top2 = res.scores.topk(2, dim=1).values
        report = {
            "workload": f"{a.videos} videos x {a.frames} frames, {len(prompts)} queries, EVA-CLIP-{a.model} {a.precision}, {world} rank(s)"
                        + (", ranks SHARING the GPU(s) over gloo: functional run, not an N-GPU result" if a.share_gpu else ""),
            "backend": a.backend,
            "n_gpus": world, "rccl_ranks": dist.get_world_size() if world > 1 else 1,
            "frames": a.videos * a.frames, "seconds_incl_input_generation": elapsed,
            "frames_per_s_incl_input_generation": a.videos * a.frames / elapsed,
            "all_ranks_hold_the_same_rows_and_ranking": bool(ranks_agree),
            "digest_key": key, **digest,
            # bit equality with the 1-rank sweep holds when every rank's block is whole encode calls of >= 64 frames (retrieval.run_corpus);
            # other shard sizes are compared by the ranks' agreement only
            "equals_committed_1rank_digest": (known[key] == digest) if (key in known and a.precision == "bf16" and whole_calls) else None,
            "shards_are_whole_calls_of_64_plus_frames": bool(whole_calls),
            "pooled_row_norm_min_max": [res.video_rows.norm(dim=1).min().item(), res.video_rows.norm(dim=1).max().item()],
            "score_min_max": [res.scores.min().item(), res.scores.max().item()],
            "median_top1_margin": (top2[:, 0] - top2[:, 1]).median().item(),
            "result_dict": {"prompts": len(res), "videos_per_prompt": len(res[prompts[0]]["videos"]),
                            "layout": "inference_video_retrieval.py:337-346"},
            "peak_memory_GB": torch.cuda.max_memory_allocated(dev) / 1e9}
        if a.rank_exact_k > 0:
            report["rank_exact"] = rank_exact
            report["equals_committed_1rank_digest"] = None           # (re-encoded rows differ from the all-bf16 digest by construction)
    if a.rank_blocks > 1 and world == 1:
        t1 = time.perf_counter()
        parts = [retrieval.corpus_block_rows(model, src, r, a.rank_blocks, a.frames) for r in range(a.rank_blocks)]
        rows = torch.cat(parts)
        again = retrieval.score_corpus(res.text_rows, rows, ids, prompts)
        _, idx2 = again.topk(10)
        torch.cuda.synchronize(dev)
        report.update({"rank_blocks": a.rank_blocks, "seconds_rank_blocks": time.perf_counter() - t1,
                       "pooled_rows_bit_identical": bool(torch.equal(rows, res.video_rows)),
                       "scores_bit_identical": bool(torch.equal(again.scores, res.scores)),
                       "top10_identical": bool(torch.equal(idx2, idx))})
    if rank == 0:
        print(json.dumps(report), flush=True)
        if a.out:
            os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
            with open(a.out, "w") as f:
                json.dump(report, f, indent=1)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0 and (not ranks_agree or report.get("equals_committed_1rank_digest") is False
                      or report.get("pooled_rows_bit_identical") is False or report.get("top10_identical") is False
                      or report.get("rank_exact", {}).get("topk_lists_equal_all_bf16x3") is False):
        raise SystemExit(1)


if __name__ == "__main__":
    main()
