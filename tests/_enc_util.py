"""Shared helpers of the encoder tests: golden fixtures (tests/golden/encoder_*.npz, written by
oracle/gen_encoder_goldens.py from the REFERENCE classes) -> config, seeded weights, batch."""
import glob
import os
from types import SimpleNamespace

import numpy as np
import torch

from oracle import encoder_ref as R

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
BATCH_KEYS = ("input_ids", "input_mask", "part2_mask", "part3_mask", "input_type_ids")


def golden_names():
    return sorted(os.path.basename(p)[len("encoder_"):-len(".npz")] for p in glob.glob(os.path.join(GOLD, "encoder_*.npz")))


def load_golden(name):
    z = np.load(os.path.join(GOLD, "encoder_%s.npz" % name), allow_pickle=False)
    cfg = {}
    for k in z.files:
        if k.startswith("cfg_"):
            v = z[k]
            cfg[k[4:]] = v.item() if v.ndim == 0 else v
    for k, v in list(cfg.items()):
        if isinstance(v, (np.str_, bytes)):
            cfg[k] = str(v)
    batch = {k: torch.from_numpy(z[k]) for k in BATCH_KEYS if k in z.files}
    extra = {k: z[k] for k in ("probe_layers", "probe_pos", "probe_hidden") if k in z.files}
    return cfg, int(z["weight_seed"]), batch, z["expected_embed"], extra


def golden_meta(name):
    """(klass, q_weight_seed or None, encode_table) of a fixture; older fixtures carry the convention in their name."""
    z = np.load(os.path.join(GOLD, "encoder_%s.npz" % name), allow_pickle=False)
    if "klass" in z.files:
        qs = int(z["q_weight_seed"])
        return str(z["klass"]), (None if qs < 0 else qs), bool(int(z["encode_table"]))
    return None, None, "question" not in name


def weights_for(cfg, seed, q_seed=None, encode_table=True):
    """(checkpoint-style state_dict incl. encoder_q.* when q_seed is given, the weights the tested forward runs on)."""
    sd = state_dict_for(cfg, seed)
    if q_seed is None:
        return sd, sd
    sdq = R.make_state_dict(cfg, q_seed)
    ck = dict(sd)
    ck.update({"encoder_q." + k[len("encoder."):]: v for k, v in sdq.items() if k.startswith("encoder.")})
    used = dict(sd)
    if not encode_table:
        used.update({k: v for k, v in sdq.items() if k.startswith("encoder.")})
    return ck, used


_SD_CACHE = {}


def state_dict_for(cfg, seed):
    key = (cfg["family"], cfg["num_hidden_layers"], cfg["vocab_size"], seed)
    if key not in _SD_CACHE:
        _SD_CACHE.clear()            # the full-size dict is ~500 MB: keep one
        _SD_CACHE[key] = R.make_state_dict(cfg, seed)
    return _SD_CACHE[key]


def hf_like_config(cfg):
    return SimpleNamespace(**{k: cfg[k] for k in ("vocab_size", "hidden_size", "num_hidden_layers", "num_attention_heads",
                                                  "intermediate_size", "max_position_embeddings", "type_vocab_size",
                                                  "pad_token_id", "layer_norm_eps")})


def ref_args(cfg, encode_table=True, fp16=False, shared_encoder=True):
    return SimpleNamespace(model_name="unused", shared_encoder=shared_encoder, no_proj=bool(cfg["no_proj"]),
                           part_pooling=str(cfg["part_pooling"]), encode_table=encode_table, fp16=fp16,
                           multi_vector=1, scheme="none")
