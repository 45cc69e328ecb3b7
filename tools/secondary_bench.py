#!/usr/bin/env python3
"""Driver-verifiable numbers for the rows of SURVEY 8 that are not the headline (BASELINE configs[3], configs[4], row f4):
the joint model's moment retrieval / moment segmentation / step captioning, its training step and the ASR sentence encoder,
each at the reference's own operating point (B = 5 = scripts/run.sh:14, T = 300 frames; beam 3 and 5, 48 words), each with
its roofline fraction and with the CPU oracle timed beside it on this host.

``bench.py`` calls ``measure()`` outside its timed region (rank 0, N = 1) and prints the result as the ``secondary`` object
of its JSON line; ``python tools/secondary_bench.py`` prints the same object on its own (profiles/rNN/secondary.json).

The reference's published speeds for these paths are val_inference_and_evaluation.ipynb:708,713,716 (T4): 38 videos/s
moment retrieval, 8 videos/s moment segmentation, 48 captions/s.

Roofline conventions: the joint model runs exact fp32 on v_mfma_f32_32x32x2_f32, dense peak 157.3 TFLOP/s
(MI355X_MICROARCH.md); algorithmic FLOPs are counted from the layer shapes below, masked / padded positions included (the
kernels compute them).  Step captioning is bound by streaming the decoder's weights once per word for the whole batch of beams
(LM head 30522 x 768 fp32 = 94 MB + 2 decoder layers): bytes per word / 8 TB/s.
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

F32_MFMA_PEAK_TFLOPS = 157.3
HBM_PEAK_GBS = 8000.0
E, H, FF, VOCAB = 512, 768, 3072, 30522


def encoder_flops_per_token(T):
    """VisualModel (module_visual.py:396-424): embedding Linear(512, 768) + 2 x (QKV, out, FFN, attention over T keys)."""
    return 2 * E * H + 2 * (2 * (4 * H * H + 2 * H * FF) + 4 * T * H)


def fusion_flops_per_token():
    """modeling.py:158-195: clip_g_map 1024->512, asr 384->512, temporal 512->512 (per token); the text map is per video."""
    return 2 * (1024 * E + 384 * E + E * E)


def _timeit(fn, reps, sync):
    """best of two groups of `reps` calls after one warm-up call (a one-off allocator growth is not the operation)"""
    fn(); sync()
    best = None
    for _ in range(2):
        t0 = time.perf_counter()
        for _ in range(reps):
            out = fn()
        sync()
        dt = (time.perf_counter() - t0) / reps
        best = dt if best is None else min(best, dt)
    return best, out


def measure(dev=None, cpu=True, log=lambda m: None):
    import hirest_amd
    from hirest_amd import synth
    from hirest_amd.sentence_encoder import SentenceTransformer
    from hirest_amd.synth import joint_inputs, train_targets, caption_targets
    dev = dev or torch.device("cuda:0")
    sync = torch.cuda.synchronize
    shapes = {k: tuple(v) for k, v in json.load(open(os.path.join(ROOT, "tests", "golden", "joint_schema.json"))).items()}
    sd = synth.joint_state_dict(shapes, 31)
    model = hirest_amd.MomentModel(n_frames=-1, asr_dim=384, args=None, clip_model=None)
    model.load_state_dict(sd, strict=False)
    model = model.to(dev).eval()
    B, T = 5, 300
    threads = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(threads)
    # the inputs of the real-reference golden joint_c300 (make_golden.py gen_joint: the REAL MomentModel.test_step at B = 5, T = 300), so the timed
    # batches' indices and boundary lists are compared with the reference's, not only with the CPU oracle's
    vis, asr, text, vis_mask, moment_mask, bounds = joint_inputs("joint.c300", B, T, 41)
    jgold5 = json.load(open(os.path.join(ROOT, "tests", "golden", "joint_predictions.json")))["c300"]
    g = lambda t: t.to(dev)
    common = {"vis_feats": g(vis), "vis_mask": g(vis_mask), "asr_feats": g(asr), "text_feat": g(text)}
    out = {"operating_point": f"B={B} videos, T={T} frames, fp32 (joint model), synthetic weights / features; CPU = oracle/ref_cpu.py, "
                              f"{threads} torch threads", "reference_T4": {"moment_retrieval_videos_per_s": 38,
                                                                           "moment_segmentation_videos_per_s": 8, "captions_per_s_beam3": 48,
                                                                           "where": "val_inference_and_evaluation.ipynb:708,713,716"}}
    if cpu:
        from oracle import ref_cpu as O

    # ---- moment retrieval (BASELINE configs[3]; modeling.py:272-308)
    log("secondary: moment retrieval")
    bmr = dict(common, tasks=["moment_retrieval"], moment_mask=g(moment_mask))
    dt, pred = _timeit(lambda: model.test_step(bmr)["prediction"], 20, sync)
    flops = B * T * (fusion_flops_per_token() + encoder_flops_per_token(T) + 2 * 2 * H)
    ent = {"value": B / dt, "unit": "videos/s", "ms_per_batch": dt * 1e3,
           "roofline": {"bound": "mfma", "achieved": flops / dt / 1e12, "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": flops / dt / 1e12 / F32_MFMA_PEAK_TFLOPS, "algorithmic_flops_per_batch": flops}}
    if cpu:
        t0 = time.perf_counter(); p_cpu, _, _ = O.moment_retrieval(sd, vis, text, asr, vis_mask, moment_mask); tc = time.perf_counter() - t0
        ent["cpu_baseline"] = {"value": B / tc, "unit": "videos/s", "cores": threads, "kind": "port", "sample": "one batch"}
        ent["indices_equal_cpu_oracle"] = bool(p_cpu == pred)
    ent["indices_equal_real_reference"] = bool(pred == jgold5["pred_moment_retrieval"])
    out["moment_retrieval"] = ent

    # ---- moment segmentation, 20 iterations (modeling.py:353-474)
    log("secondary: moment segmentation")
    bsg = dict(common, tasks=["moment_segmentation"], moment_bound_frames=bounds)
    dt, pred = _timeit(lambda: model.test_step(bsg)["prediction"], 5, sync)
    flops = B * T * (fusion_flops_per_token() + 20 * (encoder_flops_per_token(T) + 2 * H))
    ent = {"value": B / dt, "unit": "videos/s", "ms_per_batch": dt * 1e3, "iterations": 20,
           "roofline": {"bound": "mfma", "achieved": flops / dt / 1e12, "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": flops / dt / 1e12 / F32_MFMA_PEAK_TFLOPS, "algorithmic_flops_per_batch": flops}}
    if cpu:
        t0 = time.perf_counter(); s_cpu, _ = O.moment_segmentation(sd, vis, text, asr, vis_mask, bounds); tc = time.perf_counter() - t0
        ent["cpu_baseline"] = {"value": B / tc, "unit": "videos/s", "cores": threads, "kind": "port", "sample": "one batch"}
        ent["boundaries_equal_cpu_oracle"] = bool(s_cpu == pred)
    ent["boundaries_equal_real_reference"] = bool(pred == jgold5["pred_segmentation"])
    out["moment_segmentation"] = ent

    # ---- the same two tasks at the reference's DEFAULT evaluation batch (args.py:27 --eval_batch_size 32) on the inputs of the real-reference
    # golden joint_d300 (make_golden.py gen_joint: the REAL MomentModel.test_step at B = 32, T = 300): every index / boundary list compared
    log("secondary: moment retrieval / segmentation at B = 32")
    jgold = json.load(open(os.path.join(ROOT, "tests", "golden", "joint_predictions.json")))["d300"]
    B32 = jgold["B"]
    dvis, dasr, dtext, dvm, dmm, dbounds = joint_inputs("joint.d300", B32, T, 41)
    c32 = {"vis_feats": g(dvis), "vis_mask": g(dvm), "asr_feats": g(dasr), "text_feat": g(dtext)}
    dt, pred = _timeit(lambda: model.test_step(dict(c32, tasks=["moment_retrieval"], moment_mask=g(dmm)))["prediction"], 10, sync)
    flops = B32 * T * (fusion_flops_per_token() + encoder_flops_per_token(T) + 2 * 2 * H)
    out["moment_retrieval_b32"] = {"value": B32 / dt, "unit": "videos/s", "ms_per_batch": dt * 1e3, "batch": B32,
                                   "indices_equal_real_reference": bool(pred == jgold["pred_moment_retrieval"]),
                                   "roofline": {"bound": "mfma", "achieved": flops / dt / 1e12, "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                                "frac": flops / dt / 1e12 / F32_MFMA_PEAK_TFLOPS, "algorithmic_flops_per_batch": flops}}
    dt, pred = _timeit(lambda: model.test_step(dict(c32, tasks=["moment_segmentation"], moment_bound_frames=dbounds))["prediction"], 3, sync)
    flops = B32 * T * (fusion_flops_per_token() + 20 * (encoder_flops_per_token(T) + 2 * H))
    out["moment_segmentation_b32"] = {"value": B32 / dt, "unit": "videos/s", "ms_per_batch": dt * 1e3, "batch": B32, "iterations": 20,
                                      "boundaries_equal_real_reference": bool(pred == jgold["pred_segmentation"]),
                                      "roofline": {"bound": "mfma", "achieved": flops / dt / 1e12, "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                                   "frac": flops / dt / 1e12 / F32_MFMA_PEAK_TFLOPS, "algorithmic_flops_per_batch": flops}}

    # ---- the same four figures with the encoder's linear layers on split operands (MomentModel.set_precision('bf16x3'), csrc/joint_x3.hip:
    # three bf16 MFMAs per product; the reference's counterpart is autocast under --fp16, run.py:549-551).  Rooflines against the dense bf16
    # peak / 3 on the same ALGORITHMIC flops; indices / boundary lists compared with the real reference's.
    log("secondary: moment retrieval / segmentation, bf16x3 encoder")
    model.set_precision("bf16x3")
    try:
        for key, batch, n, reps, gold, iters in (
                ("moment_retrieval", bmr, B, 20, jgold5["pred_moment_retrieval"], 1),
                ("moment_segmentation", bsg, B, 5, jgold5["pred_segmentation"], 20),
                ("moment_retrieval_b32", dict(c32, tasks=["moment_retrieval"], moment_mask=g(dmm)), B32, 10, jgold["pred_moment_retrieval"], 1),
                ("moment_segmentation_b32", dict(c32, tasks=["moment_segmentation"], moment_bound_frames=dbounds), B32, 3, jgold["pred_segmentation"], 20)):
            dt, pred = _timeit(lambda: model.test_step(batch)["prediction"], reps, sync)
            flops = n * T * (fusion_flops_per_token() + iters * (encoder_flops_per_token(T) + (2 * 2 * H if iters == 1 else 2 * H)))
            out[key + "_bf16x3"] = {"value": n / dt, "unit": "videos/s", "ms_per_batch": dt * 1e3, "batch": n, "iterations": iters,
                                    "speedup_vs_fp32": (n / dt) / out[key]["value"],
                                    "predictions_equal_real_reference": bool(pred == gold),
                                    "roofline": {"bound": "mfma", "achieved": flops / dt / 1e12, "peak": 2500.0 / 3, "unit": "TFLOP/s",
                                                 "frac": flops / dt / 1e12 / (2500.0 / 3), "algorithmic_flops_per_batch": flops,
                                                 "note": "peak = dense bf16 MFMA peak / 3: every product of the encoder's linear layers costs three bf16 MFMAs"}}
    finally:
        model.set_precision("fp32")

    # ---- step captioning (BASELINE configs[4]; modeling.py:556-632) at its own operating point: B = 5, 15-frame
    # moments -> 20 trimmed frames, 48 words, on the inputs of the real-reference goldens tests/golden/caption_predictions.json
    # cases c3 / c5 (make_golden.py gen_caption: the REAL MomentModel.test_step), so ALL FIVE captions of the timed batch are
    # compared with the reference's token ids
    from hirest_amd.synth import CAPTION_CASES
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "caption_predictions.json")))
    sep_bias = dict(model.named_parameters())["clip4cap_model.decoder.classifier.cls.predictions.bias"]
    sep_saved = sep_bias.detach().clone()
    with torch.no_grad():
        sep_bias[102] += 1.5                                             # as the golden's checkpoint: [SEP] reachable
    layer_w = (3 * H * H + H * H) + (H * H + H * H) + 2 * H * FF       # self qkv + out, cross q + out, FFN (cross K/V are per batch)
    bytes_per_word = 4 * (2 * layer_w + H * H + VOCAB * H)              # fp32 weights streamed once per word for all beams
    for beams in (3, 5):
        log(f"secondary: step captioning, beam {beams}")
        case = f"c{beams}"
        cB, cT, cbeams, lens = CAPTION_CASES[case]
        assert (cB, cT, cbeams) == (B, T, beams)
        cvis, casr, ctext, cvm, _, _ = joint_inputs(f"cap.{case}", cB, cT, 47)
        cmm = torch.zeros(cB, cT, dtype=torch.long)
        for b in range(cB):
            cmm[b, 5 + b:5 + b + lens[b]] = 1
        bcp = {"tasks": ["step_captioning"], "vis_feats": g(cvis), "vis_mask": g(cvm), "asr_feats": g(casr), "text_feat": g(ctext),
               "moment_mask": cmm}
        dt, r = _timeit(lambda: model.test_step(bcp, num_beams=beams, return_ids=True), 3, sync)
        words = max(len(h) for h in r["token_ids"])
        gbs = bytes_per_word * 48 / dt / 1e9
        want = [[int(t) for t in p.split()] for p in gold[case]["prediction"]]
        ent = {"value": B / dt, "unit": "captions/s", "ms_per_batch": dt * 1e3, "beam": beams, "max_words": 48,
               "longest_hypothesis_words": words,
               "token_ids_equal_real_reference": f"{sum(int(list(a) == b) for a, b in zip(r['token_ids'], want))} of {len(want)} captions",
               "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                            "algorithmic_bytes_per_word_step": bytes_per_word,
                            "note": "decoder + LM-head fp32 weights once per word for the whole batch of beams, 48 word steps"}}
        if cpu and beams == 5:
            sdc = dict(sd)
            sdc["clip4cap_model.decoder.classifier.cls.predictions.bias"] = sd["clip4cap_model.decoder.classifier.cls.predictions.bias"].clone()
            sdc["clip4cap_model.decoder.classifier.cls.predictions.bias"][102] += 1.5
            t0 = time.perf_counter()
            c_cpu, _ = O.step_captioning(sdc, cvis[:1], ctext[:1], casr[:1], cmm[:1], beams=beams)
            tc = time.perf_counter() - t0
            ent["cpu_baseline"] = {"value": 1 / tc, "unit": "captions/s", "cores": threads, "kind": "port", "sample": "one caption (video 0)"}
            ent["token_ids_equal_cpu_oracle_on_sample"] = bool(list(c_cpu[0]) == list(r["token_ids"][0]))
        out[f"step_captioning_beam{beams}"] = ent
        # the same batch as an evaluation loop sees it (run.py:328-336: one test_step per loader batch): MomentModel.caption_batches
        # captions consecutive loader batches with ONE beam search over the union of their beam rows (up to 160 rows: 12 batches of 5
        # videos = two searches of 30 videos at beam 5, one after the other; beam 3: 50 + 10), so a word's decoder + LM-head weights are
        # streamed once per SEARCH, not once per batch.  Same token ids per batch (tests/test_gpu_joint.py).
        nb, ns = 12, 1
        many = [bcp] * nb
        model.caption_batches(many, num_beams=beams, streams=ns)                           # warm-up: streams, allocator pools, graphs
        sync(); t0 = time.perf_counter()
        res = model.caption_batches(many, num_beams=beams, streams=ns, return_ids=True)
        sync(); dtp = (time.perf_counter() - t0) / nb
        per_search = max(1, model.CAPTION_ROWS_IN_FLIGHT // beams // B)                     # loader batches per merged search
        searches = -(-nb // per_search)
        gbs = bytes_per_word * 48 * searches / (dtp * nb) / 1e9
        # actual rows per search (beam 3: 10 + 2 batches = 150 + 30 rows, not 2 x 150): FLOPs are counted on the rows that exist
        rows = [min(per_search, nb - i * per_search) * B * beams for i in range(searches)]
        tfl = 2.0 * (bytes_per_word / 4) * sum(rows) * 48 / (dtp * nb) / 1e12
        out[f"step_captioning_beam{beams}_pipelined"] = {
            "value": B / dtp, "unit": "captions/s", "ms_per_batch": dtp * 1e3, "beam": beams, "max_words": 48, "batches": nb,
            "merged_searches": searches, "beam_rows_per_search": rows, "searches_in_flight": ns,
            "how": "MomentModel.caption_batches (default merge=True): loader batches of 5 videos captioned by one beam search per 160 beam rows; "
                   "one search after the other (a second search in flight gains nothing at ~150 rows)",
            "token_ids_equal_real_reference": f"{sum(int(list(a) == b) for r_ in res for a, b in zip(r_['token_ids'], want))} of {nb * len(want)} captions",
            "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                         "algorithmic_bytes_per_word_step": bytes_per_word,
                         "note": "the decoder + LM-head weights ONCE per word per merged search (not per loader batch); at 90 - 150 rows the word "
                                 "is matrix-pipe time, not weight-stream time: see mfma_f32",
                         "mfma_f32": {"achieved": tfl, "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tfl / F32_MFMA_PEAK_TFLOPS,
                                      "note": "2 x weights x beam rows per word (attention and the tail not counted), exact-fp32 MFMA"}}}
        # the reference's DEFAULT evaluation batch (args.py:27 --eval_batch_size 32): one test_step on the inputs of the real-reference
        # goldens d3 / d5 (make_golden.py: the REAL MomentModel.test_step at B = 32), all 32 captions compared
        case = f"d{beams}"
        dB, dT, _, dlens = CAPTION_CASES[case]
        dvis, dasr, dtext, dvm, _, _ = joint_inputs(f"cap.{case}", dB, dT, 47)
        dmm = torch.zeros(dB, dT, dtype=torch.long)
        for b in range(dB):
            dmm[b, 5 + b:5 + b + dlens[b]] = 1
        bd = {"tasks": ["step_captioning"], "vis_feats": g(dvis), "vis_mask": g(dvm), "asr_feats": g(dasr), "text_feat": g(dtext), "moment_mask": dmm}
        dtd, rd = _timeit(lambda: model.test_step(bd, num_beams=beams, return_ids=True), 3, sync)
        wantd = [[int(t) for t in p.split()] for p in gold[case]["prediction"]]
        gbs = bytes_per_word * 48 / dtd / 1e9
        tfl = 2.0 * (bytes_per_word / 4) * dB * beams * 48 / dtd / 1e12
        out[f"step_captioning_beam{beams}_b32"] = {
            "value": dB / dtd, "unit": "captions/s", "ms_per_batch": dtd * 1e3, "beam": beams, "max_words": 48, "batch": dB,
            "beam_rows": dB * beams, "how": "MomentModel.test_step at the reference's default --eval_batch_size 32 (args.py:27)",
            "token_ids_equal_real_reference": f"{sum(int(list(a) == b) for a, b in zip(rd['token_ids'], wantd))} of {len(wantd)} captions",
            "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                         "algorithmic_bytes_per_word_step": bytes_per_word,
                         "note": "one weight stream per word for all 96 / 160 beam rows (48 word steps: no sample of the golden batch ends early "
                                 "enough to stop the search); the word is matrix-pipe time at this size: see mfma_f32",
                         "mfma_f32": {"achieved": tfl, "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tfl / F32_MFMA_PEAK_TFLOPS,
                                      "note": "2 x weights x beam rows per word (attention and the tail not counted), exact-fp32 MFMA"}}}
        # the same B = 32 batch with the encoder pass and the LM head (>= 64 beam rows) on split operands (MomentModel.set_precision('bf16x3'))
        model.set_precision("bf16x3")
        try:
            dtx, rx = _timeit(lambda: model.test_step(bd, num_beams=beams, return_ids=True), 3, sync)
        finally:
            model.set_precision("fp32")
        out[f"step_captioning_beam{beams}_b32_bf16x3"] = {
            "value": dB / dtx, "unit": "captions/s", "ms_per_batch": dtx * 1e3, "beam": beams, "max_words": 48, "batch": dB, "beam_rows": dB * beams,
            "speedup_vs_fp32": dtd / dtx,
            "how": "as step_captioning_beamN_b32, LM head of every word (and the encoder pass) as three bf16 MFMAs per product; decoder layers exact fp32",
            "token_ids_equal_real_reference": f"{sum(int(list(a) == b) for a, b in zip(rx['token_ids'], wantd))} of {len(wantd)} captions"}
    with torch.no_grad():
        sep_bias.copy_(sep_saved)

    # ---- joint-model training step (row f4; run.py:238-295): train_step + backward + clip_grad_norm_ + AdamW
    log("secondary: training step")
    model.train()
    opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=1e-5)
    st, et, seg, prev = train_targets(f"tb.{T}", B, T, 61, bounds)
    pin = lambda t: t.pin_memory()                    # the reference's loaders deliver pinned batches (hirest_dataset.py:614,624)
    btr = {"vis_feats": pin(vis), "vis_mask": pin(vis_mask), "asr_feats": pin(asr), "text_feat": pin(text), "tasks": ["moment_retrieval"],
           "moment_mask": pin(moment_mask), "moment_retrieval_start_target": pin(st), "moment_retrieval_end_target": pin(et)}

    def step():
        opt.zero_grad(set_to_none=True)
        loss = model.train_step(btr)["loss"]
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        opt.step()
        return loss
    dt, _ = _timeit(step, 10, sync)
    flops = 3 * B * T * (fusion_flops_per_token() + encoder_flops_per_token(T))      # forward + dX + dW
    ent = {"value": dt * 1e3, "unit": "ms/step", "higher_is_better": False, "videos_per_s": B / dt, "task": "moment_retrieval",
           "includes": "train_step + backward + clip_grad_norm_ + AdamW over the 63 M trainable parameters",
           "roofline": {"bound": "mfma", "achieved": flops / dt / 1e12, "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": flops / dt / 1e12 / F32_MFMA_PEAK_TFLOPS, "algorithmic_flops_per_step": flops}}
    try:      # the same loop with torch.optim.AdamW(fused=True) (the caller's option; trainer_base.py:56 uses the class' defaults)
        opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=1e-5, fused=True)
        dtf, _ = _timeit(step, 10, sync)
        ent["ms_per_step_with_fused_adamw"] = dtf * 1e3
    except Exception as e:      # noqa: BLE001 (an optimizer option this torch build may not have)
        ent["ms_per_step_with_fused_adamw"] = None
    if cpu:
        psd = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point()}
        t0 = time.perf_counter()
        O.moment_retrieval_loss(psd, vis, text, asr, vis_mask, moment_mask, st, et).backward()
        tc = time.perf_counter() - t0
        ent["cpu_baseline"] = {"value": tc * 1e3, "unit": "ms/step", "cores": threads, "kind": "port",
                               "sample": "oracle loss under torch autograd, forward + backward only (no optimizer)"}
    out["train_step"] = ent
    # the same loop with the encoder blocks' forward and dX products on split operands (MomentModel.set_precision('bf16x3') covers training:
    # csrc/train_block.hip precision 1; gradient goldens held by tests/test_gpu_train.py at the fp32 path's bars)
    model.set_precision("bf16x3")
    try:
        opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=1e-5)
        dx3, _ = _timeit(step, 10, sync)
        ent3 = {"value": dx3 * 1e3, "unit": "ms/step", "higher_is_better": False, "videos_per_s": B / dx3, "task": "moment_retrieval",
                "speedup_vs_fp32": dt / dx3,
                "how": "encoder blocks: forward and dX products as three bf16 MFMAs per product (weights split once per step, one grouped launch); "
                       "dW, attention, LayerNorm, GELU, losses exact fp32",
                "roofline": {"bound": "mfma", "achieved": flops / dx3 / 1e12, "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                             "frac": flops / dx3 / 1e12 / F32_MFMA_PEAK_TFLOPS, "algorithmic_flops_per_step": flops,
                             "note": "priced against the exact-fp32 MFMA peak like train_step (dW and attention still run there)"}}
        try:
            opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=1e-5, fused=True)
            dx3f, _ = _timeit(step, 10, sync)
            ent3["ms_per_step_with_fused_adamw"] = dx3f * 1e3
        except Exception:      # noqa: BLE001
            ent3["ms_per_step_with_fused_adamw"] = None
        out["train_step_bf16x3"] = ent3
    finally:
        model.set_precision("fp32")
    model.eval()
    del opt

    # ---- ASR sentence encoder (row f4 tail; extract_ASR_embedding.py:14,25,54), MiniLM-L6 schema
    log("secondary: ASR sentence encoder")
    cfg = synth.MINILM_L6
    bsd = synth.bert_state_dict(cfg, 52)
    rows = synth.sentence_ids("asr_bench", 2048, 7, cfg["vocab_size"], 4, 40)
    enc = SentenceTransformer(config=cfg, state_dict=bsd).eval().to(dev)
    dt, emb = _timeit(lambda: enc.encode_ids(rows), 3, sync)
    toks = sum(map(len, rows))
    D, I, L = cfg["hidden_size"], cfg["intermediate_size"], cfg["num_hidden_layers"]
    flops = L * (toks * 2 * (4 * D * D + 2 * D * I) + sum(4 * len(r) * len(r) * D for r in rows))
    ent = {"value": len(rows) / dt, "unit": "sentences/s", "tokens_per_s": toks / dt, "sentences": len(rows), "tokens": toks,
           "roofline": {"bound": "mfma", "achieved": flops / dt / 1e12, "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": flops / dt / 1e12 / F32_MFMA_PEAK_TFLOPS, "algorithmic_flops_per_call": flops}}
    if cpu:
        n = 64
        t0 = time.perf_counter(); ref = O.sentence_embeddings(bsd, rows[:n], cfg["num_attention_heads"]); tc = time.perf_counter() - t0
        ent["cpu_baseline"] = {"value": n / tc, "unit": "sentences/s", "cores": threads, "kind": "port", "sample": f"first {n} sentences"}
        ent["max_abs_diff_vs_cpu_oracle_on_sample"] = (emb[:n].cpu() - ref).abs().max().item()
    out["asr_sentence_encoder"] = ent
    return out


if __name__ == "__main__":
    res = measure(cpu="--no-cpu" not in sys.argv, log=lambda m: print(m, file=sys.stderr, flush=True))
    print(json.dumps(res, indent=1))
