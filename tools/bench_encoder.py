#!/usr/bin/env python3
"""tools/bench_encoder.py — corpus-encode throughput (blocks/s) of the HIP encoder on one GPU.
Synthetic inputs per SURVEY.md §8d: ids uniform in [3, 50264], <s> at 0, </s> at 128,129,511,
full-length mask, part2 = [1,128), part3 = [130,511); roberta-base shape, random weights (seed 99)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import encoder_ref as R            # weight/shape generator only (not timed)
from otter_amd import encoder as E
from types import SimpleNamespace


def synth_batch(B, L, seed=777, vocab=50265, ragged=False):
    if ragged:
        return ragged_batch(B, L, seed, vocab)
    rs = np.random.RandomState(seed)
    ids = rs.randint(3, vocab, size=(B, L)).astype(np.int64)
    ids[:, 0] = 0
    if L >= 512:
        ids[:, 128] = 2; ids[:, 129] = 2; ids[:, L - 1] = 2
    mask = np.ones((B, L), np.int64)
    p2 = np.zeros((B, L), np.int64); p2[:, 1:min(128, L)] = 1
    p3 = np.zeros((B, L), np.int64); p3[:, min(130, L - 1):L - 1] = 1
    return {k: torch.from_numpy(v).cuda() for k, v in
            dict(input_ids=ids, input_mask=mask, part2_mask=p2, part3_mask=p3).items()}


def ragged_batch(B, L, seed=777, vocab=50265):
    """SURVEY.md §8d ragged variant: lengths ~U[64, L], padded to L the way the reference's tokenizer call
    does (padding='max_length', retrieval/data/tr_dataset.py:139-141); '<s> A </s></s> B </s> <pad>*'."""
    rs = np.random.RandomState(seed)
    ids = np.full((B, L), 1, np.int64)
    mask = np.zeros((B, L), np.int64)
    p2 = np.zeros((B, L), np.int64)
    p3 = np.zeros((B, L), np.int64)
    for b in range(B):
        n = int(rs.randint(64, L + 1))
        a = max(2, n // 4)
        ids[b, :n] = rs.randint(3, vocab, size=n)
        ids[b, 0] = 0
        ids[b, a] = 2; ids[b, a + 1] = 2; ids[b, n - 1] = 2
        mask[b, :n] = 1
        p2[b, 1:a] = 1
        p3[b, a + 2:n - 1] = 1
    return {k: torch.from_numpy(v).cuda() for k, v in
            dict(input_ids=ids, input_mask=mask, part2_mask=p2, part3_mask=p3).items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ragged", action="store_true", help="lengths ~U[64, len] padded to len (SURVEY.md §8d)")
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--len", type=int, default=512)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--layers", type=int, default=12)
    a = ap.parse_args()
    cfg = R.make_config("roberta", layers=a.layers)
    sd = R.make_state_dict(cfg, 99)
    hf = SimpleNamespace(**{k: cfg[k] for k in ("vocab_size", "hidden_size", "num_hidden_layers", "num_attention_heads",
                                                "intermediate_size", "max_position_embeddings", "type_vocab_size",
                                                "pad_token_id", "layer_norm_eps")})
    args = SimpleNamespace(shared_encoder=True, no_proj=True, part_pooling="first", encode_table=True)
    model = E.RobertaSingleEncoderThreeCatPool(hf, args, dtype=a.dtype)
    model.load_state_dict(sd, strict=False)
    model.to("cuda:0")
    batch = synth_batch(a.batch, a.len, ragged=a.ragged)
    for _ in range(a.warmup):
        out = model(batch)["embed"]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = model(batch)["embed"]
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    L, H = a.len, 768
    flops = a.layers * (24.0 * L * H * H + 4.0 * L * L * H) * a.batch       # of the padded shape
    rec = {"blocks_per_s": a.batch / dt, "ms_per_batch": dt * 1e3, "batch": a.batch, "len": L,
           "dtype": a.dtype, "tflops": flops / dt / 1e12, "finite": bool(torch.isfinite(out).all()),
           "ragged": a.ragged, "packed": os.environ.get("OTTER_NO_PACK") != "1",
           "ln_fused": os.environ.get("OTTER_NO_LN_FUSE") != "1"}
    if a.ragged:
        rec["mean_len"] = float(batch["input_mask"].sum().item()) / a.batch
        rec["tflops_note"] = "flops of the padded shape; the packed layout skips the padded rows"
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
