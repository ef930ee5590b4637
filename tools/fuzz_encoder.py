"""tools/fuzz_encoder.py [--seconds S] [--seed N] — randomized parity run of the HIP encoder against the CPU restatement.

Random family (roberta / bert), pooling, projection, one-vector or three-part class, 1-2 layers, batch 1 ... 24,
sequence length 8 ... 512 (any value, not only multiples of a tile) and ragged lengths.  fp32 path: every embedding
within 1e-4 + 1e-4 |ref| of oracle/encoder_ref.py (fp32 torch on the CPU; 'sum' pooling 1e-3 absolute, as in
tests/test_encoder_gpu.py).  bf16 path: cosine >= 0.999 per 768-vector against the same reference.  Exit status 1
if any case fails.  The oracle is the checker here, never the path.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from oracle import encoder_ref as R  # noqa: E402
from tests import _enc_util as U  # noqa: E402
from tests.test_encoder_gpu import _model  # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=240.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--cases", type=int, default=0)
    args = ap.parse_args(argv)
    rng = np.random.RandomState(args.seed)
    t_end = time.time() + args.seconds
    bad = case = 0
    while time.time() < t_end and bad < 5 and (args.cases <= 0 or case < args.cases):
        case += 1
        family = str(rng.choice(["roberta", "roberta", "bert"]))
        three = rng.rand() < 0.8
        pooling = str(rng.choice(["first", "mean", "sum", "max", "att_self_pool"])) if three else "first"
        cfg = R.make_config(family, layers=int(rng.randint(1, 3)), vocab_size=2000, three_cat=bool(three),
                            no_proj=bool(rng.rand() < 0.6), part_pooling=pooling)
        B = int(rng.randint(1, 25))
        L = int(rng.choice([int(rng.randint(8, 513)), int(rng.choice([64, 128, 256, 384, 512]))]))
        wseed, bseed = int(rng.randint(1, 1000)), int(rng.randint(1, 1000))
        batch = R.make_batch(cfg, B, L, bseed, ragged=bool(rng.rand() < 0.8))
        sd = U.state_dict_for(cfg, wseed)
        with torch.no_grad():
            exp = R.EncoderRef(cfg, sd)(batch)["embed"].numpy()
        dev = {k: v.cuda() for k, v in batch.items()}
        H = cfg["hidden_size"]
        msg = []
        got = _model(cfg, wseed)(dev)["embed"].cpu().numpy()
        err = np.abs(got - exp)
        atol = 1e-3 if pooling == "sum" else 1e-4
        if got.shape != exp.shape or not (err <= atol + 1e-4 * np.abs(exp)).all():
            msg.append("fp32 max err %g" % float(err.max()))
        got16 = _model(cfg, wseed, dtype="bf16")(dev)["embed"].cpu().numpy()
        cosmin = 1.0
        for part in range(exp.shape[1] // H):
            g, e = got16[:, part * H:(part + 1) * H], exp[:, part * H:(part + 1) * H]
            cos = (g * e).sum(1) / (np.linalg.norm(g, axis=1) * np.linalg.norm(e, axis=1))
            cosmin = min(cosmin, float(cos.min()))
        if not (np.isfinite(got16).all() and cosmin >= 0.999):
            msg.append("bf16 min cosine %g" % cosmin)
        print("case %4d  %-7s %-13s three=%d proj=%d layers=%d B=%2d L=%3d  fp32 err %.2e  bf16 cos %.5f  %s" % (
            case, family, pooling, three, not cfg["no_proj"], cfg["num_hidden_layers"], B, L, float(err.max()), cosmin,
            "ok" if not msg else "FAILED: " + "; ".join(msg) + " (weights %d, batch %d)" % (wseed, bseed)), flush=True)
        bad += 1 if msg else 0
    print("cases: %d, failed: %d" % (case, bad), flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
