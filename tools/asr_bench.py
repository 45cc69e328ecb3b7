#!/usr/bin/env python3
"""Throughput of the ASR sentence encoder (MiniLM-L6 schema, synthetic weights) on subtitle-like input, with the CPU oracle timed
beside it on a sample.   python tools/asr_bench.py [--sentences 2048] [--max-len 40]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hirest_amd import synth  # noqa: E402
from hirest_amd.sentence_encoder import SentenceTransformer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sentences", type=int, default=2048)
    ap.add_argument("--max-len", type=int, default=40)
    ap.add_argument("--cpu-sample", type=int, default=64)
    a = ap.parse_args()
    cfg = synth.MINILM_L6
    sd = synth.bert_state_dict(cfg, 52)
    rows = synth.sentence_ids("asr_bench", a.sentences, 7, cfg["vocab_size"], 4, a.max_len)
    dev = torch.device("cuda:0")
    m = SentenceTransformer(config=cfg, state_dict=sd).eval().to(dev)
    m.encode_ids(rows[:64]); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); out = m.encode_ids(rows); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    toks = sum(map(len, rows))
    print(f"GPU: {a.sentences} sentences ({toks} tokens, lengths {min(map(len, rows))}..{max(map(len, rows))}) in {best * 1e3:.1f} ms = "
          f"{a.sentences / best:.0f} sentences/s, {toks / best / 1e3:.0f} k tokens/s")
    from oracle import ref_cpu
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    t0 = time.perf_counter(); ref = ref_cpu.sentence_embeddings(sd, rows[:a.cpu_sample], cfg["num_attention_heads"]); dt = time.perf_counter() - t0
    print(f"CPU oracle ({torch.get_num_threads()} threads): {a.cpu_sample / dt:.0f} sentences/s; max |GPU - CPU| on the sample "
          f"{(out[:a.cpu_sample].cpu() - ref).abs().max().item():.2e}")


if __name__ == "__main__":
    main()
