"""Synthetic encoder inputs: shape dicts, seeded weights and batches of the reference's layout.

Used by bench.py's encode leg, __graft_entry__.smoke(), the tests and the oracle (which re-exports
these names); plain numpy/torch data generation, no arithmetic of the path itself.
Shapes follow HF's RobertaConfig / BertConfig as the reference instantiates them
(retrieval/models/tb_retriever.py:151-163) and the batch dict of em_collate_bert
(retrieval/data/encode_datasets.py:393-417).
"""
from collections import OrderedDict

import numpy as np
import torch


def make_config(family="roberta", layers=12, vocab_size=None, hidden=768, heads=12, inter=3072,
                three_cat=True, no_proj=True, part_pooling="first"):
    """Shape/config dict with the fields of HF's RobertaConfig/BertConfig that the forward reads."""
    if family == "roberta":
        cfg = dict(vocab_size=vocab_size or 50265, max_position_embeddings=514, type_vocab_size=1,
                   pad_token_id=1, layer_norm_eps=1e-5, position_mode="roberta")
    elif family == "bert":
        cfg = dict(vocab_size=vocab_size or 30522, max_position_embeddings=512, type_vocab_size=2,
                   pad_token_id=0, layer_norm_eps=1e-12, position_mode="bert")
    else:
        raise ValueError(family)
    cfg.update(family=family, hidden_size=hidden, num_hidden_layers=layers, num_attention_heads=heads,
               intermediate_size=inter, three_cat=three_cat, no_proj=no_proj, part_pooling=part_pooling)
    return cfg


def make_state_dict(cfg, seed):
    """Seeded weights under the reference wrapper's state_dict key names (utils.py:12-28 loads
    exactly these): 'encoder.embeddings.*', 'encoder.encoder.layer.N.*', 'encoder.pooler.*',
    'project.0.*' (Linear), 'project.1.*' (LayerNorm).  numpy RandomState (frozen legacy stream);
    LayerNorm gamma/beta are non-trivial on purpose (HF's default 1/0 hides scaling bugs)."""
    rs = np.random.RandomState(seed)
    H, I = cfg["hidden_size"], cfg["intermediate_size"]

    def nrm(shape, std):
        return torch.from_numpy((rs.standard_normal(size=shape) * std).astype(np.float32))

    def ln(prefix, sd):
        sd[prefix + ".weight"] = 1.0 + nrm((H,), 0.1)
        sd[prefix + ".bias"] = nrm((H,), 0.1)

    sd = OrderedDict()
    e = "encoder.embeddings."
    sd[e + "word_embeddings.weight"] = nrm((cfg["vocab_size"], H), 0.05)
    sd[e + "position_embeddings.weight"] = nrm((cfg["max_position_embeddings"], H), 0.05)
    sd[e + "token_type_embeddings.weight"] = nrm((cfg["type_vocab_size"], H), 0.05)
    ln(e + "LayerNorm", sd)
    for i in range(cfg["num_hidden_layers"]):
        p = "encoder.encoder.layer.%d." % i
        for name in ("attention.self.query", "attention.self.key", "attention.self.value",
                     "attention.output.dense"):
            sd[p + name + ".weight"] = nrm((H, H), 0.04)
            sd[p + name + ".bias"] = nrm((H,), 0.02)
        ln(p + "attention.output.LayerNorm", sd)
        sd[p + "intermediate.dense.weight"] = nrm((I, H), 0.04)
        sd[p + "intermediate.dense.bias"] = nrm((I,), 0.02)
        sd[p + "output.dense.weight"] = nrm((H, I), 0.03)
        sd[p + "output.dense.bias"] = nrm((H,), 0.02)
        ln(p + "output.LayerNorm", sd)
    sd["encoder.pooler.dense.weight"] = nrm((H, H), 0.04)   # present in checkpoints, unused by the path
    sd["encoder.pooler.dense.bias"] = nrm((H,), 0.02)
    sd["project.0.weight"] = nrm((H, H), 0.04)
    sd["project.0.bias"] = nrm((H,), 0.02)
    ln("project.1", sd)
    if str(cfg.get("part_pooling", ""))[:3] == "att":
        # ...ThreeCatAtt: Linear(H, 1) per side (tb_retriever.py:419-421); drawn last, so every other
        # configuration keeps its stream
        for side in ("tab", "psg"):
            sd["self_proj_%s.weight" % side] = nrm((1, H), 0.3)
            sd["self_proj_%s.bias" % side] = nrm((1,), 0.1)
    return sd


def make_batch(cfg, B, L, seed, ragged=True, question=False):
    """Synthetic batch dict in the layout of em_collate_bert (retrieval/data/encode_datasets.py:393-417):
    int64 [B, L] input_ids / input_mask / part2_mask / part3_mask (+ input_type_ids for BERT), built
    by the mask rule of retrieval/data/tr_dataset.py:117-153 on a fake '<s> A </s></s> B </s> <pad>*'
    token layout (question=True: single segment, part3 = part2, :125-133)."""
    rs = np.random.RandomState(seed)
    pad = cfg["pad_token_id"]
    rob = cfg["family"] == "roberta"
    cls_id, sep_id = (0, 2) if rob else (101, 102)
    lo = 3 if rob else 1000
    ids = np.full((B, L), pad, dtype=np.int64)
    mask = np.zeros((B, L), dtype=np.int64)
    p2 = np.zeros((B, L), dtype=np.int64)
    p3 = np.zeros((B, L), dtype=np.int64)
    tt = np.zeros((B, L), dtype=np.int64)
    for b in range(B):
        n = L if (not ragged or b == 0) else int(rs.randint(max(8, L // 4), L + 1))
        toks = rs.randint(lo, cfg["vocab_size"], size=n)
        toks[0] = cls_id
        if question:
            toks[n - 1] = sep_id
            p2[b, 1:n - 1] = 1
            p3[b] = p2[b]
        else:
            a_end = int(rs.randint(2, max(3, n // 3)))          # first separator
            toks[a_end] = sep_id
            nsep = 2 if rob else 1                               # roberta: </s></s>
            if rob:
                toks[a_end + 1] = sep_id
            toks[n - 1] = sep_id
            p2[b, 1:a_end] = 1
            p3[b, a_end + nsep:n - 1] = 1
            tt[b, a_end + 1:n] = 0 if rob else 1
        ids[b, :n] = toks
        mask[b, :n] = 1
    batch = {"input_ids": torch.from_numpy(ids), "input_mask": torch.from_numpy(mask),
             "part2_mask": torch.from_numpy(p2), "part3_mask": torch.from_numpy(p3)}
    if not rob:
        batch["input_type_ids"] = torch.from_numpy(tt)
    return batch
