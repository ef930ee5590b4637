#!/usr/bin/env python3
"""tools/exp_predict_gap.py — where the encode leg's wall time per batch exceeds the device time of a forward:
predict() over N host batches against N forwards of a resident batch, and the host time of each pipeline step."""
import os
import sys
import time
from types import SimpleNamespace

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import encoder_ref as R            # weights / shapes only
from otter_amd import encoder as E
from otter_amd import encode_driver as ED
from tools.bench_encoder import synth_batch


def main():
    cfg = R.make_config("roberta", layers=12)
    sd = R.make_state_dict(cfg, 99)
    hf = SimpleNamespace(**{k: cfg[k] for k in ("vocab_size", "hidden_size", "num_hidden_layers", "num_attention_heads",
                                                "intermediate_size", "max_position_embeddings", "type_vocab_size",
                                                "pad_token_id", "layer_norm_eps")})
    args = SimpleNamespace(shared_encoder=True, no_proj=True, part_pooling="first", encode_table=True)
    model = E.RobertaSingleEncoderThreeCatPool(hf, args, dtype="bf16")
    model.load_state_dict(sd, strict=False)
    model.to("cuda:0")
    b = synth_batch(128, 512)
    hbs = [{k: v.cpu() for k, v in synth_batch(128, 512, seed=700 + i).items()} for i in range(4)]
    for _ in range(3):
        model(b)
    ED.predict(model, hbs[:2])
    torch.cuda.synchronize()
    for rep in range(3):
        t0 = time.perf_counter()
        ED.predict(model, [hbs[i % 4] for i in range(20)])
        torch.cuda.synchronize()
        print("repeat %d: predict of 20 batches %.3f ms/batch" % (rep, (time.perf_counter() - t0) / 20 * 1e3), flush=True)
    for n in (20, 60):
        t0 = time.perf_counter()
        ED.predict(model, [hbs[i % 4] for i in range(n)])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n * 1e3
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        t0 = time.perf_counter()
        for _ in range(n):
            model(b)
        host_launch = (time.perf_counter() - t0) / n * 1e3
        e1.record()
        torch.cuda.synchronize()
        print("batches %d: predict %.3f ms/batch, resident forwards %.3f ms/batch (host enqueue %.3f ms/forward)"
              % (n, dt, e0.elapsed_time(e1) / n, host_launch), flush=True)
    # predict with device-resident inputs (no H2D), and without D2H
    for to_cpu in (True, False):
        t0 = time.perf_counter()
        ED.predict(model, [b] * 20, to_cpu=to_cpu)
        torch.cuda.synchronize()
        print("resident inputs, to_cpu=%s: %.3f ms/batch" % (to_cpu, (time.perf_counter() - t0) / 20 * 1e3), flush=True)
    t0 = time.perf_counter()
    ED.predict(model, [hbs[i % 4] for i in range(20)], to_cpu=False)
    torch.cuda.synchronize()
    print("host inputs, to_cpu=False: %.3f ms/batch" % ((time.perf_counter() - t0) / 20 * 1e3), flush=True)


if __name__ == "__main__":
    main()
