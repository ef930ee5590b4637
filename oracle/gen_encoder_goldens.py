#!/usr/bin/env python3
"""oracle/gen_encoder_goldens.py — run the REFERENCE encoder classes on seeded weights/inputs and
commit their outputs as fixtures (tests/golden/encoder_*.npz).

Runs only in the build container (needs /root/reference; the reference cannot travel).  The
reference is imported as it is; the two modules it imports but does not ship are stubbed on a
temporary sys.path entry OUTSIDE the reference tree:
    attention/SelfAttention.py   (retrieval/models/tb_retriever.py:20; only the *Att classes use it)
    pathos/multiprocessing.py    (retrieval/data/tr_dataset.py:23; unused)
Pretrained weights are not available offline, so each case builds a local HF model directory
from oracle.encoder_ref.make_state_dict(cfg, seed) and points args.model_name at it; the
checkpoint goes through the reference's own load_saved (retrieval/utils/utils.py:12-28).

Every case also checks oracle.encoder_ref.EncoderRef (the restatement that travels to the GPU
box) against the reference output before writing the fixture.

    python oracle/gen_encoder_goldens.py            # regenerate all fixtures
"""
import os
import sys
import tempfile
from types import SimpleNamespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import encoder_ref as R  # noqa: E402

REF = "/root/reference"
GOLD = os.path.join(ROOT, "tests", "golden")


def _stubs(tmp):
    os.makedirs(os.path.join(tmp, "attention"))
    os.makedirs(os.path.join(tmp, "pathos"))
    open(os.path.join(tmp, "attention", "__init__.py"), "w").close()
    with open(os.path.join(tmp, "attention", "SelfAttention.py"), "w") as f:
        f.write("class ScaledDotProductAttention:\n    def __init__(self, *a, **k):\n"
                "        raise RuntimeError('stub: not shipped with the reference')\n")
    open(os.path.join(tmp, "pathos", "__init__.py"), "w").close()
    with open(os.path.join(tmp, "pathos", "multiprocessing.py"), "w") as f:
        f.write("from multiprocessing import Pool\n")
    sys.path.insert(0, tmp)
    sys.path.insert(0, REF)


def _hf_model_dir(cfg, sd, tmp, tag):
    from transformers import BertConfig, BertModel, RobertaConfig, RobertaModel
    kw = dict(vocab_size=cfg["vocab_size"], hidden_size=cfg["hidden_size"],
              num_hidden_layers=cfg["num_hidden_layers"], num_attention_heads=cfg["num_attention_heads"],
              intermediate_size=cfg["intermediate_size"], max_position_embeddings=cfg["max_position_embeddings"],
              type_vocab_size=cfg["type_vocab_size"], layer_norm_eps=cfg["layer_norm_eps"],
              pad_token_id=cfg["pad_token_id"], hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
    if cfg["family"] == "roberta":
        hcfg, cls = RobertaConfig(**kw), RobertaModel
    else:
        hcfg, cls = BertConfig(**kw), BertModel
    m = cls(hcfg)
    inner = {k[len("encoder."):]: v for k, v in sd.items() if k.startswith("encoder.")}
    missing, unexpected = m.load_state_dict(inner, strict=False)
    assert not unexpected, unexpected
    assert all("position_ids" in k or "token_type_ids" in k for k in missing), missing
    d = os.path.join(tmp, "hf_" + tag)
    m.save_pretrained(d)
    return d, hcfg


ONLY = [t for t in os.environ.get("OTTER_GOLDEN_ONLY", "").split(",") if t]   # regenerate just these tags


def run_case(tag, cfg, wseed, batch, tmp, klass="threecat", encode_table=True, probes=None, dup_b1=False, q_seed=None):
    """klass: 'threecat' / 'onevec' / 'ctx' (the ...CtxEncoder classes).  q_seed: shared_encoder=False with a second,
    different set of encoder weights under 'encoder_q.*' in the checkpoint (tb_retriever.py:160-161,205-218)."""
    if ONLY and tag not in ONLY:
        return
    from retrieval.models.retriever import CtxEncoder, RobertaCtxEncoder, RobertaSingleEncoder, SingleEncoder
    from retrieval.models.tb_retriever import RobertaSingleEncoderThreeCatPool, SingleEncoderThreeCatPool
    from retrieval.models.tb_retriever import RobertaSingleEncoderThreeCatAtt, SingleEncoderThreeCatAtt
    from retrieval.utils.utils import load_saved
    sd = R.make_state_dict(cfg, wseed)
    mdir, hcfg = _hf_model_dir(cfg, sd, tmp, tag)
    args = SimpleNamespace(model_name=mdir, shared_encoder=q_seed is None, no_proj=cfg["no_proj"],
                           part_pooling=cfg["part_pooling"], encode_table=encode_table, multi_vector=1, scheme="none")
    rob = cfg["family"] == "roberta"
    ck_sd = dict(sd)
    sd_used = sd                      # the weights the forward under test actually runs on
    if q_seed is not None:
        sdq = R.make_state_dict(cfg, q_seed)
        for k, v in sdq.items():
            if k.startswith("encoder."):
                ck_sd["encoder_q." + k[len("encoder."):]] = v
        if not encode_table:
            sd_used = dict(sd)
            sd_used.update({k: v for k, v in sdq.items() if k.startswith("encoder.")})
    if klass == "ctx" and not rob:    # CtxEncoder names its HF model `encoder_c`
        ck_sd = {("encoder_c." + k[len("encoder."):] if k.startswith("encoder.") else k): v for k, v in ck_sd.items()}
    if klass == "threecat" and str(cfg["part_pooling"])[:3] == "att":
        model = (RobertaSingleEncoderThreeCatAtt if rob else SingleEncoderThreeCatAtt)(hcfg, args)
    elif klass == "threecat":
        model = (RobertaSingleEncoderThreeCatPool if rob else SingleEncoderThreeCatPool)(hcfg, args)
    elif klass == "ctx":
        model = (RobertaCtxEncoder if rob else CtxEncoder)(hcfg, args)
    else:
        model = (RobertaSingleEncoder if rob else SingleEncoder)(hcfg, args)
    ck = os.path.join(tmp, tag + ".pt")
    torch.save({"module." + k: v for k, v in ck_sd.items()}, ck)   # DataParallel-style prefix: load_saved strips it
    model = load_saved(model, ck, exact=False)
    model.eval()
    run_batch = batch
    if dup_b1:   # the reference cannot run B=1 with 'first' pooling (tb_retriever.py:55 squeezes the batch dim)
        run_batch = {k: torch.cat([v, v], 0) for k, v in batch.items()}
    with torch.no_grad():
        if rob or klass == "ctx":
            out = model(run_batch)["embed"]
        else:
            # SingleEncoder*.forward reads self.encode_table, which only the Roberta subclasses set
            # (SURVEY.md §7): call encode_seq directly, as the survey prescribes for BERT
            if klass == "threecat":
                out = model.encode_seq(run_batch["input_ids"], run_batch["input_mask"], run_batch["input_type_ids"],
                                       part2_mask=run_batch["part2_mask"], part3_mask=run_batch["part3_mask"])
            else:
                out = model.encode_seq(run_batch["input_ids"], run_batch["input_mask"], run_batch["input_type_ids"])
        hs = None
        if probes is not None:
            hs = model.encoder(input_ids=run_batch["input_ids"], attention_mask=run_batch["input_mask"],
                               token_type_ids=run_batch.get("input_type_ids"), output_hidden_states=True).hidden_states
    if dup_b1:
        assert torch.equal(out[0], out[1])
        out = out[:1]
    out = out.float().numpy()
    # the restatement that travels must reproduce the reference here
    mine = R.EncoderRef(cfg, sd_used)(batch)["embed"].numpy()
    err = float(np.abs(mine - out).max())
    scale = float(np.abs(out).max())
    print("%-34s embed %-12s max|ref| %.3f  max|restatement - reference| %.3g" % (tag, out.shape, scale, err))
    assert err <= 2e-5 * max(1.0, scale), "restatement diverges from the reference: %g" % err
    save = {k: v.numpy() for k, v in batch.items()}
    save["expected_embed"] = out
    save["weight_seed"] = np.int64(wseed)
    save["q_weight_seed"] = np.int64(-1 if q_seed is None else q_seed)
    save["klass"] = np.asarray(klass)
    save["encode_table"] = np.int64(1 if encode_table else 0)
    if probes is not None:
        layers, pos = probes
        save["probe_layers"] = np.asarray(layers, dtype=np.int64)
        save["probe_pos"] = np.asarray(pos, dtype=np.int64)
        save["probe_hidden"] = np.stack([hs[l][:, pos, :].float().numpy() for l in layers])  # [nl, B, np, H]
        ref_h = R.EncoderRef(cfg, sd)
        for li, l in enumerate(layers):
            mh = ref_h.hidden(batch["input_ids"], batch["input_mask"], batch.get("input_type_ids"), l)
            e2 = float(np.abs(mh[:, pos, :].numpy() - save["probe_hidden"][li]).max())
            assert e2 <= 2e-5 * max(1.0, float(np.abs(save["probe_hidden"][li]).max())), (l, e2)
    cfg_items = {("cfg_" + k): np.asarray(v) for k, v in cfg.items()}
    np.savez_compressed(os.path.join(GOLD, "encoder_%s.npz" % tag), **save, **cfg_items)


def main():
    os.makedirs(GOLD, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    with tempfile.TemporaryDirectory() as tmp:
        _stubs(tmp)
        small = dict(layers=2, vocab_size=1000)
        # default configuration (three_cat, no_proj, 'first'), ragged batch, table blocks
        cfg = R.make_config("roberta", **small)
        run_case("rob2_first", cfg, 11, R.make_batch(cfg, 4, 96, 21), tmp, probes=([0, 1, 2], [0, 1, 7]))
        # every pooling method of pooling_masked_part
        for pool in ("mean", "sum", "max", "cls"):
            cfg = R.make_config("roberta", part_pooling=pool, **small)
            run_case("rob2_" + pool, cfg, 11, R.make_batch(cfg, 4, 96, 21), tmp)
        # projection head on
        cfg = R.make_config("roberta", no_proj=False, **small)
        run_case("rob2_first_proj", cfg, 12, R.make_batch(cfg, 3, 80, 22), tmp)
        # questions: single segment, L = 70, part3 = part2, encode_q path
        cfg = R.make_config("roberta", **small)
        run_case("rob2_question", cfg, 11, R.make_batch(cfg, 5, 70, 23, question=True), tmp, encode_table=False)
        # empty part masks fall back to row 0 under 'first'
        b = R.make_batch(cfg, 3, 64, 24)
        b["part3_mask"][1] = 0
        b["part2_mask"][2] = 0
        run_case("rob2_first_emptymask", cfg, 11, b, tmp)
        # B = 1 (expected value from the duplicated batch)
        run_case("rob2_first_b1", cfg, 11, R.make_batch(cfg, 1, 64, 25), tmp, dup_b1=True)
        # one-vector encoders of retrieval/models/retriever.py
        cfg = R.make_config("roberta", three_cat=False, **small)
        run_case("rob2_onevec", cfg, 13, R.make_batch(cfg, 4, 96, 26), tmp, klass="onevec")
        cfg = R.make_config("roberta", three_cat=False, no_proj=False, **small)
        run_case("rob2_onevec_proj", cfg, 13, R.make_batch(cfg, 4, 96, 26), tmp, klass="onevec")
        # BERT family: arange positions, token types, eps 1e-12
        cfg = R.make_config("bert", layers=2, vocab_size=2000)
        run_case("bert2_first", cfg, 14, R.make_batch(cfg, 4, 96, 27), tmp)
        cfg = R.make_config("bert", layers=2, vocab_size=2000, part_pooling="mean", no_proj=False)
        run_case("bert2_mean_proj", cfg, 14, R.make_batch(cfg, 4, 96, 27), tmp)
        # attention pooling of the ...ThreeCatAtt encoders (scripts/encode_corpus.py:50-66), one row with an empty part
        cfg = R.make_config("roberta", part_pooling="att_self_pool", **small)
        b = R.make_batch(cfg, 4, 96, 28)
        b["part3_mask"][2] = 0
        run_case("rob2_att_self_pool", cfg, 15, b, tmp)
        cfg = R.make_config("bert", layers=2, vocab_size=2000, part_pooling="att_self_pool", no_proj=False)
        run_case("bert2_att_self_pool_proj", cfg, 16, R.make_batch(cfg, 3, 80, 29), tmp)
        # full roberta-base shape: 12 layers, vocab 50265, L = 512, ragged
        cfg = R.make_config("roberta")
        run_case("rob12_first_L512", cfg, 99, R.make_batch(cfg, 2, 512, 777), tmp,
                 probes=([0, 1, 6, 12], [0, 1, 100]))
        run_case("rob12_question_L70", cfg, 99, R.make_batch(cfg, 3, 70, 778, question=True), tmp,
                 encode_table=False)
        # the corpus-side one-vector classes of retrieval/models/retriever.py:513-564
        cfg = R.make_config("roberta", three_cat=False, no_proj=False, **small)
        run_case("rob2_ctx", cfg, 17, R.make_batch(cfg, 4, 96, 30), tmp, klass="ctx")
        cfg = R.make_config("bert", layers=2, vocab_size=2000, three_cat=False, no_proj=True)
        run_case("bert2_ctx", cfg, 18, R.make_batch(cfg, 3, 80, 31), tmp, klass="ctx")
        # shared_encoder=False: questions run on encoder_q.* (different weights), blocks on encoder.*
        cfg = R.make_config("roberta", **small)
        run_case("rob2_question_notshared", cfg, 11, R.make_batch(cfg, 5, 70, 23, question=True), tmp,
                 encode_table=False, q_seed=19)
        run_case("rob2_first_notshared", cfg, 11, R.make_batch(cfg, 4, 96, 21), tmp, q_seed=19)


if __name__ == "__main__":
    main()
