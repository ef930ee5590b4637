"""oracle/encoder_ref.py — CPU restatement (plain torch, fp32/fp64) of OTTeR's bi-encoder forward.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke(), bench.py's cpu_baseline
leg and oracle/gen_encoder_goldens.py; never by otter_amd/.

Pinned: oracle/gen_encoder_goldens.py runs the REFERENCE classes themselves in the build
container (imported from /root/reference with the two absent modules stubbed)
on the same seeded weights and inputs, checks this restatement against them, and commits
the reference's outputs under tests/golden/encoder_*.npz.  tests/test_encoder_oracle.py
re-checks this file against those fixtures on every run.

What is restated (reference file:line):
  * pooling_masked_part                      retrieval/models/tb_retriever.py:42-58
  * SingleEncoderThreeCatPool.encode_seq     retrieval/models/tb_retriever.py:172-195
    (= encode_q with shared_encoder,         :197-227; Roberta subclass forward :256-261)
  * SingleEncoder.encode_seq (one vector)    retrieval/models/retriever.py:270-286, :340-345
  * the HF AutoModel forward it calls (:180-185): RoBERTa/BERT-base post-LN encoder,
    transformers/models/roberta/modeling_roberta.py (embeddings :56-155 incl. the
    cumulative-sum position ids :142-155; self-attention :157-184; layer :186-399) and
    the BERT equivalents (arange positions, token types).
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

POOLS = ("mean", "sum", "max", "cls", "first")


from otter_amd.synthetic import make_batch, make_config, make_state_dict  # noqa: E402,F401  (data generators live in the package)


def att_self_pool(hidden, mask, weight, bias):
    """One side of tb_retriever.py:510-523: softmax over positions of Linear(H,1)(h) + (1 - m) * -1e6,
    weighted sum of h (`.squeeze()` of the B = 1 case not reproduced, as in pooling_masked_part below)."""
    m = mask.to(hidden.dtype).unsqueeze(2)
    logits = torch.nn.functional.linear(hidden, weight, bias)
    att = torch.softmax(logits + (1 - m) * -1e6, dim=1)
    return torch.sum(att * hidden, dim=1)


def pooling_masked_part(hidden, mask, method="mean"):
    """tb_retriever.py:42-58, with the B=1 `.squeeze()` accident removed (the reference then
    fails in torch.cat; the build returns [1, H])."""
    maskf = mask.to(hidden.dtype)
    if method == "mean":
        return (hidden * maskf.unsqueeze(2)).sum(1) / maskf.sum(1, keepdim=True)
    if method == "sum":
        return (hidden * maskf.unsqueeze(2)).sum(1)
    if method == "max":
        return (hidden + ((1 - maskf) * -1e5).unsqueeze(2)).max(dim=1)[0]
    if method == "first":
        L = mask.shape[1]
        tmp = mask * torch.arange(L, 0, -1, device=mask.device)
        index = torch.argmax(tmp, 1)                      # first 1 of the mask; 0 for an empty mask
        return hidden[torch.arange(hidden.shape[0]), index]
    return hidden[:, 0, :]                                # 'cls' and anything else


class EncoderRef:
    """Functional restatement; `dtype` torch.float32 (parity reference) or float64 (arbiter)."""

    def __init__(self, cfg, state_dict, dtype=torch.float32):
        self.cfg = dict(cfg)
        self.dtype = dtype
        self.w = {k: v.to(dtype) for k, v in state_dict.items()}

    # -- HF embeddings (modeling_roberta.py:56-155 / modeling_bert.py)
    def embed(self, ids, type_ids=None):
        c, w = self.cfg, self.w
        e = "encoder.embeddings."
        if c["position_mode"] == "roberta":
            m = (ids != c["pad_token_id"]).long()
            pos = torch.cumsum(m, dim=1) * m + c["pad_token_id"]
        else:
            pos = torch.arange(ids.shape[1]).unsqueeze(0).expand_as(ids)
        if type_ids is None:
            type_ids = torch.zeros_like(ids)
        x = w[e + "word_embeddings.weight"][ids] + w[e + "position_embeddings.weight"][pos] + \
            w[e + "token_type_embeddings.weight"][type_ids]
        return torch.nn.functional.layer_norm(x, (c["hidden_size"],), w[e + "LayerNorm.weight"],
                                              w[e + "LayerNorm.bias"], c["layer_norm_eps"])

    def layer(self, x, i, key_bias):
        c, w = self.cfg, self.w
        p = "encoder.encoder.layer.%d." % i
        B, L, H = x.shape
        nh = c["num_attention_heads"]
        dh = H // nh
        lin = torch.nn.functional.linear

        def heads(t):
            return t.view(B, L, nh, dh).permute(0, 2, 1, 3)

        q = heads(lin(x, w[p + "attention.self.query.weight"], w[p + "attention.self.query.bias"]))
        k = heads(lin(x, w[p + "attention.self.key.weight"], w[p + "attention.self.key.bias"]))
        v = heads(lin(x, w[p + "attention.self.value.weight"], w[p + "attention.self.value.bias"]))
        s = q @ k.transpose(-1, -2) * (dh ** -0.5) + key_bias           # [B, nh, L, L]
        a = torch.softmax(s, dim=-1) @ v
        a = a.permute(0, 2, 1, 3).reshape(B, L, H)
        a = lin(a, w[p + "attention.output.dense.weight"], w[p + "attention.output.dense.bias"])
        x = torch.nn.functional.layer_norm(a + x, (H,), w[p + "attention.output.LayerNorm.weight"],
                                           w[p + "attention.output.LayerNorm.bias"], c["layer_norm_eps"])
        f = lin(x, w[p + "intermediate.dense.weight"], w[p + "intermediate.dense.bias"])
        f = torch.nn.functional.gelu(f)                                  # exact erf form
        f = lin(f, w[p + "output.dense.weight"], w[p + "output.dense.bias"])
        return torch.nn.functional.layer_norm(f + x, (H,), w[p + "output.LayerNorm.weight"],
                                              w[p + "output.LayerNorm.bias"], c["layer_norm_eps"])

    def hidden(self, ids, mask, type_ids=None, n_layers=None):
        """last_hidden_state after n_layers layers (None = all), [B, L, H]."""
        n = self.cfg["num_hidden_layers"] if n_layers is None else n_layers
        x = self.embed(ids, type_ids)
        key_bias = ((1 - mask.to(self.dtype)) * torch.finfo(self.dtype).min)[:, None, None, :]
        for i in range(n):
            x = self.layer(x, i, key_bias)
        return x

    def forward(self, batch):
        """model(batch) -> {'embed': [B, D]} (tb_retriever.py:256-261 / retriever.py:340-345)."""
        c, w = self.cfg, self.w
        h = self.hidden(batch["input_ids"], batch["input_mask"], batch.get("input_type_ids"))
        part1 = h[:, 0, :]
        if not c["no_proj"]:
            part1 = torch.nn.functional.linear(part1, w["project.0.weight"], w["project.0.bias"])
            part1 = torch.nn.functional.layer_norm(part1, (c["hidden_size"],), w["project.1.weight"],
                                                   w["project.1.bias"], c["layer_norm_eps"])
        if not c["three_cat"]:
            return {"embed": part1}
        if c["part_pooling"] == "att_self_pool":
            part2 = att_self_pool(h, batch["part2_mask"], w["self_proj_tab.weight"], w["self_proj_tab.bias"])
            part3 = att_self_pool(h, batch["part3_mask"], w["self_proj_psg.weight"], w["self_proj_psg.bias"])
            return {"embed": torch.cat([part1, part2, part3], dim=1)}
        part2 = pooling_masked_part(h, batch["part2_mask"], c["part_pooling"])
        part3 = pooling_masked_part(h, batch["part3_mask"], c["part_pooling"])
        return {"embed": torch.cat([part1, part2, part3], dim=1)}

    __call__ = forward
