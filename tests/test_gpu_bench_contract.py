"""bench.py's output contract (one JSON line with the metric, the live roofline of the dominant kernel and the CPU
baseline) on a reduced batch, so that a kernel or profile-format change cannot silently break what the driver parses."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_contract():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "1", "--warmup", "1", "--frames", "256",
                        "--chunk", "256", "--cpu-frames", "1"], capture_output=True, text=True, timeout=600, cwd=REPO)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["metric"].startswith("encoded frames/sec") and d["unit"] == "frames/s" and d["higher_is_better"] is True
    assert d["n_gpus"] == 1 and d["rccl_ranks"] == 1 and d["steps"] == 1 and d["warmup"] == 1 and d["scaling"] == "weak" and d["dtype"] == "bf16"
    assert d["data"] == "synthetic" and d["vs_baseline"] is None and "workload" in d["config"]
    assert d["value"] > 100 and abs(d["value"] - 256 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    rf = d["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and rf["peak"] == 2500.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and 0.05 < rf["frac"] < 1.0
    assert "traffic" in rf and rf["kernel"].startswith("gemm<") and rf["avg_launch_ms"] > 0
    # achieved = algorithmic flops of the dominant kernel / its live average launch duration
    assert abs(rf["achieved"] - rf["algorithmic_flops_per_launch"] / (rf["avg_launch_ms"] * 1e-3) / 1e12) < 1e-6 * rf["achieved"]
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["unit"] == "frames/s" and cb["value"] > 0 and cb["cores"] >= 1 and cb["sample"]
    assert cb["min_cosine_gpu_vs_cpu_on_sample"] > 0.999
    assert cb["single_thread"] > 0 and str(cb["cores"]) in cb["by_threads"] and cb["host_cpus"] >= cb["cores"]
    assert set(cb["by_threads"]) >= {"1", str(min(32, cb["host_cpus"]))} and all(v["frames_per_s"] > 0 for v in cb["by_threads"].values())
    assert all(v["how"].startswith(("measured", "extrapolated from a 2-block probe")) for v in cb["by_threads"].values())
    # 64 / 128 threads bound close (OMP_PROC_BIND=close OMP_PLACES=cores, child processes) are probed before 32 is settled on
    assert all(f"{t}_pinned" in cb["by_threads"] for t in (64, 128) if t < cb["host_cpus"])
    # matched R@k at EVA-CLIP-g/14 scale against the real reference's rankings (tests/golden/eva_g14_c3.npz)
    mr = d["matched_recall"]
    assert mr["queries"] == 546 and mr["videos"] == 256
    assert mr["matched_R@5"] == 100.0 and mr["matched_R@10"] == 100.0 and mr["matched_R@1"] >= 85.0
    assert mr["top1_exact_where_margin_gt_2x_error"] is True and mr["pooled_min_cosine_vs_reference"] > 0.999
    # the reference-precision towers (precision='fp32') reproduce the reference's ranks; their speed is a separate figure
    p32 = mr["precision_fp32"]
    assert p32["top1_flips"] == 0 and p32["matched_R@1"] == 100.0 and p32["max_abs_score_error"] < 2e-5 and p32["frames_per_s"] > 10
    # precision='bf16x3': the reference's ranks from split-operand products, with its own roofline line (peak / 3: three MFMAs per product)
    x3 = mr["precision_bf16x3"]
    assert x3["top1_flips"] == 0 and x3["matched_R@1"] == 100.0 and x3["max_abs_score_error"] < 5e-6 and x3["top10_lists_identical"] >= 540
    assert x3["frames_per_s"] > 2 * p32["frames_per_s"]
    assert abs(x3["roofline"]["peak"] - 2500.0 / 3) < 1e-6 and 0.2 < x3["roofline"]["frac"] < 1.0
    assert abs(x3["roofline"]["frac"] - x3["roofline"]["achieved"] / x3["roofline"]["peak"]) < 1e-12
    assert rf["traffic_missing"] == (rf["traffic"] is None)
    # margin-guarded re-rank (retrieval.rerank_exact): the reference's top-1 on the pinned corpus, and a throughput leg on a synthetic corpus
    rx = mr["rank_exact_rerank_k10"]
    assert rx["top1_flips"] == 0 and rx["matched_R@1"] == 100.0 and rx["top10_lists_identical"] >= 540 and 0 < rx["reencoded_fraction"] <= 1.0
    rt = mr["rank_exact_throughput"]
    assert rt["rank_exact_effective_frames_per_s"] == rt["k1"]["effective_frames_per_s"] > x3["frames_per_s"] and rt["k1"]["reencoded_fraction"] < 0.5
    # executed vs unpruned work (the last block serves x[:, 0] only): the tower fraction is priced on executed FLOPs
    assert rf["executed_gflop_per_frame"] < rf["unpruned_gflop_per_frame"] == 534.06
    assert abs(rf["whole_tower_frac"] - d["value"] * rf["executed_gflop_per_frame"] / 1e3 / 2500.0) < 1e-9
    # the other SURVEY-8 rows at the reference's operating point: value + roofline + CPU oracle each
    sec = d["secondary"]
    units = {"moment_retrieval": "videos/s", "moment_segmentation": "videos/s", "step_captioning_beam3": "captions/s",
             "step_captioning_beam5": "captions/s", "train_step": "ms/step", "asr_sentence_encoder": "sentences/s"}
    for key, unit in units.items():
        e = sec[key]
        assert e["unit"] == unit and e["value"] > 0, key
        r_ = e["roofline"]
        assert r_["bound"] in ("mfma", "hbm") and abs(r_["frac"] - r_["achieved"] / r_["peak"]) < 1e-12 and 0 < r_["frac"] < 1, key
        if key != "step_captioning_beam3":
            assert e["cpu_baseline"]["kind"] == "port" and e["cpu_baseline"]["value"] > 0 and e["cpu_baseline"]["cores"] >= 1, key
    assert sec["moment_retrieval"]["indices_equal_cpu_oracle"] and sec["moment_segmentation"]["boundaries_equal_cpu_oracle"]
    # the timed B = 5, T = 300 batch is the real-reference golden joint_c300: exact indices / boundary lists of the REAL MomentModel
    assert sec["moment_retrieval"]["indices_equal_real_reference"] and sec["moment_segmentation"]["boundaries_equal_real_reference"]
    # the reference's default --eval_batch_size 32 (args.py:27): indices and boundary lists of the REAL MomentModel (joint_predictions.json d300)
    assert sec["moment_retrieval_b32"]["indices_equal_real_reference"] and sec["moment_segmentation_b32"]["boundaries_equal_real_reference"]
    assert sec["moment_retrieval_b32"]["value"] > sec["moment_retrieval"]["value"]
    assert sec["step_captioning_beam5"]["token_ids_equal_cpu_oracle_on_sample"]
    for beams in (3, 5):           # all five captions of the timed batch equal the REAL reference's (caption_predictions.json c3 / c5)
        assert sec[f"step_captioning_beam{beams}"]["token_ids_equal_real_reference"] == "5 of 5 captions"
        pl = sec[f"step_captioning_beam{beams}_pipelined"]            # twelve loader batches of it: merged beam searches of up to 160 rows
        assert pl["unit"] == "captions/s" and pl["merged_searches"] >= 2 and pl["token_ids_equal_real_reference"] == "60 of 60 captions"
        assert pl["value"] > 1.5 * sec[f"step_captioning_beam{beams}"]["value"]
        b32 = sec[f"step_captioning_beam{beams}_b32"]                 # the reference's default eval batch: all 32 captions = the REAL reference's
        assert b32["token_ids_equal_real_reference"] == "32 of 32 captions" and b32["beam_rows"] == 32 * beams
        assert b32["value"] > 1.5 * sec[f"step_captioning_beam{beams}"]["value"] and 0 < b32["roofline"]["mfma_f32"]["frac"] < 1
    assert sec["moment_retrieval"]["value"] > 38 and sec["moment_segmentation"]["value"] > 8 and sec["step_captioning_beam3"]["value"] > 48
    # the joint model's split-operand precision (MomentModel.set_precision('bf16x3'), round 6): the real reference's indices / boundary lists /
    # token ids, faster than the fp32 path on the same batch
    for k in ("moment_retrieval", "moment_segmentation", "moment_retrieval_b32", "moment_segmentation_b32"):
        x = sec[k + "_bf16x3"]
        assert x["predictions_equal_real_reference"] is True and x["value"] > sec[k]["value"] and abs(x["roofline"]["peak"] - 2500.0 / 3) < 1e-6, k
    for beams in (3, 5):
        x = sec[f"step_captioning_beam{beams}_b32_bf16x3"]
        assert x["token_ids_equal_real_reference"] == "32 of 32 captions" and x["value"] > sec[f"step_captioning_beam{beams}_b32"]["value"]
    tx = sec["train_step_bf16x3"]          # the training step with the encoder blocks' forward / dX products on split operands (csrc/train_block.hip)
    # (no speed gate against train_step: both legs are paced by the host's enqueue rate, which moves by +- 10 % between two runs on one box)
    assert tx["unit"] == "ms/step" and 0 < tx["value"] < 2 * sec["train_step"]["value"] and tx["ms_per_step_with_fused_adamw"] > 0


def test_bench_gpus_flag_is_binding():
    """`python bench.py --gpus N` on a box with fewer than N GPUs must fail, not report a 1-GPU run;
    with >= 2 GPUs it must print n_gpus = rccl_ranks = 2 from a self-launched 2-rank RCCL job."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--frames", "64",
                        "--chunk", "64", "--no-cpu-baseline", "--no-matched-recall"], capture_output=True, text=True, timeout=900, cwd=REPO)
    if torch.cuda.device_count() < 2:
        assert r.returncode != 0 and "GPU(s) are visible" in r.stderr
        assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
    else:
        assert r.returncode == 0, r.stderr[-2000:]
        d = json.loads([l for l in r.stdout.strip().splitlines() if l.startswith("{")][0])
        assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["config"]["global_batch"] == 128


def test_bench_rejects_partial_videos():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--frames", "100"], capture_output=True, text=True, timeout=300, cwd=REPO)
    assert r.returncode != 0 and "multiple of 32" in r.stderr


def test_bench_two_rank_path_runs_on_one_gpu_over_gloo():
    """The N-rank code path of bench.py — self-launch under torch.distributed.run, per-rank model and frames, timed_steps' barrier + MAX
    over ranks, RowGather of the pooled rows, replicated ranking, rank-0 JSON line — as two processes sharing this box's GPU over gloo
    (RCCL refuses two ranks on one device).  Functional only: the line says so (INVALID) and its value is not a 2-GPU result."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-gpu", "--steps", "2", "--warmup", "1",
                        "--frames", "64", "--chunk", "64", "--no-cpu-baseline", "--no-matched-recall", "--no-secondary"],
                       capture_output=True, text=True, timeout=900, cwd=REPO, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1                                           # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["config"]["global_batch"] == 128 and d["config"]["parallelism"] == "dp2" and d["value"] > 0
    assert abs(d["value"] - 2 * 64 * 2 / (d["ms_per_step"] * 2 / 1e3)) < 1e-6 * d["value"]     # whole-job frames / max-over-ranks time
    assert "functional run" in d["INVALID"] and d["roofline"]["launches"] > 0
