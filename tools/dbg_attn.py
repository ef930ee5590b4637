#!/usr/bin/env python3
"""tools/dbg_attn.py — localise attention errors: bf16 vs fp32 hidden states after ONE layer, per 32-row block."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace
from oracle import encoder_ref as R
from otter_amd import encoder as E

L = int(sys.argv[1]) if len(sys.argv) > 1 else 512
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cfg = R.make_config("roberta", layers=1)
sd = R.make_state_dict(cfg, 99)
hf = SimpleNamespace(**{k: cfg[k] for k in ("vocab_size", "hidden_size", "num_hidden_layers", "num_attention_heads",
                                            "intermediate_size", "max_position_embeddings", "type_vocab_size",
                                            "pad_token_id", "layer_norm_eps")})
args = SimpleNamespace(shared_encoder=True, no_proj=True, part_pooling="first", encode_table=True)
rs = np.random.RandomState(5)
ids = torch.from_numpy(rs.randint(3, 50265, size=(B, L)).astype(np.int64)).cuda()
mask = torch.ones(B, L, dtype=torch.int64).cuda()
hs = {}
for dt in ("fp32", "bf16"):
    m = E.RobertaSingleEncoderThreeCatPool(hf, args, dtype=dt)
    m.load_state_dict(sd, strict=False)
    m.to("cuda:0")
    hs[dt] = m.hidden_states(ids, mask, n_layers=1).float().cpu().numpy()
d = np.abs(hs["fp32"] - hs["bf16"])
print("overall max %.4f mean %.5f" % (d.max(), d.mean()))
for b in range(B):
    print("b", b, " ".join("%.3f" % d[b, i:i + 32].max() for i in range(0, L, 32)))
