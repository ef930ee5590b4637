"""GPU parity of the HIP encoder (through the C ABI) against the fixtures produced by the
REFERENCE classes (tests/golden/encoder_*.npz) and against the CPU restatement.

fp32 path: |embed - reference| <= 1e-4 absolute + 1e-4 relative, on embeddings of
magnitude <= ~5 (sum pooling: ~200).  bf16 path (reported separately, as north_star says):
cosine >= 0.999 per row against the fp32 reference.
"""
import numpy as np
import pytest
import torch

from oracle import encoder_ref as R
from tests import _enc_util as U

pytestmark = pytest.mark.gpu

CLASSES = {}


def _model(cfg, seed, dtype="fp32", encode_table=True, klass=None, q_seed=None):
    from otter_amd import encoder as E
    att = cfg["three_cat"] and str(cfg["part_pooling"])[:3] == "att"
    if klass == "ctx":
        # CtxEncoder keeps its HF model under `encoder_c`: hand it a checkpoint with the reference's key names
        rob = cfg["family"] == "roberta"
        sd = U.state_dict_for(cfg, seed)
        if not rob:
            sd = {("encoder_c." + k[len("encoder."):] if k.startswith("encoder.") else k): v for k, v in sd.items()}
        m = (E.RobertaCtxEncoder if rob else E.CtxEncoder)(U.hf_like_config(cfg), U.ref_args(cfg), dtype=dtype)
        missing, unexpected = m.load_state_dict(sd, strict=False)
        assert not missing, missing
        return m.to("cuda:0").eval()
    if q_seed is not None:
        cls = E.RobertaSingleEncoderThreeCatPool if cfg["family"] == "roberta" else E.SingleEncoderThreeCatPool
        m = cls(U.hf_like_config(cfg), U.ref_args(cfg, encode_table=encode_table, shared_encoder=False), dtype=dtype)
        missing, unexpected = m.load_state_dict(U.weights_for(cfg, seed, q_seed, encode_table)[0], strict=False)
        assert not missing and not unexpected, (missing, unexpected)
        return m.to("cuda:0").eval()
    if cfg["family"] == "roberta":
        cls = (E.RobertaSingleEncoderThreeCatAtt if att else E.RobertaSingleEncoderThreeCatPool) if cfg["three_cat"] \
            else E.RobertaSingleEncoder
    else:
        cls = (E.SingleEncoderThreeCatAtt if att else E.SingleEncoderThreeCatPool) if cfg["three_cat"] else E.SingleEncoder
    m = cls(U.hf_like_config(cfg), U.ref_args(cfg, encode_table=encode_table), dtype=dtype)
    m.load_state_dict(U.state_dict_for(cfg, seed), strict=False)
    return m.to("cuda:0").eval()


def _close(got, exp, rtol=1e-4, atol=1e-4):
    err = np.abs(got - exp)
    bad = err > atol + rtol * np.abs(exp)
    assert not bad.any(), "max abs err %g at %r (expected %g)" % (err.max(), np.unravel_index(err.argmax(), err.shape),
                                                               exp.flat[err.argmax()])


@pytest.mark.parametrize("name", U.golden_names())
def test_fp32_forward_matches_reference_fixture(name):
    cfg, seed, batch, expected, extra = U.load_golden(name)
    klass, q_seed, encode_table = U.golden_meta(name)
    model = _model(cfg, seed, encode_table=encode_table, klass=klass, q_seed=q_seed)
    out = model({k: v.cuda() for k, v in batch.items()})["embed"]
    assert out.is_cuda and out.dtype == torch.float32 and tuple(out.shape) == expected.shape
    got = out.cpu().numpy()
    if cfg["part_pooling"] == "sum":
        _close(got, expected, rtol=1e-4, atol=1e-3)      # sums of up to 96 rows
    else:
        _close(got, expected)
    if extra:
        for li, l in enumerate(extra["probe_layers"]):
            h = model.hidden_states(batch["input_ids"], batch["input_mask"], batch.get("input_type_ids"), int(l))
            h = h[:, torch.as_tensor(extra["probe_pos"]).cuda(), :].cpu().numpy()
            _close(h, extra["probe_hidden"][li])


@pytest.mark.parametrize("name", ["rob2_first", "rob2_mean", "bert2_first", "rob12_first_L512", "rob2_att_self_pool"])
def test_bf16_forward_close_to_fp32_reference(name):
    cfg, seed, batch, expected, _ = U.load_golden(name)
    model = _model(cfg, seed, dtype="bf16")
    got = model({k: v.cuda() for k, v in batch.items()})["embed"].cpu().numpy()
    H = cfg["hidden_size"]
    for part in range(expected.shape[1] // H):
        g, e = got[:, part * H:(part + 1) * H], expected[:, part * H:(part + 1) * H]
        cos = (g * e).sum(1) / (np.linalg.norm(g, axis=1) * np.linalg.norm(e, axis=1))
        assert (cos >= 0.999).all(), (name, part, cos)


def test_full_hidden_state_matches_restatement():
    # every row of the hidden state (padding rows included) against the CPU restatement
    cfg = R.make_config("roberta", layers=2, vocab_size=1000)
    sd = U.state_dict_for(cfg, 11)
    batch = R.make_batch(cfg, 3, 200, 31)
    ref = R.EncoderRef(cfg, sd)
    model = _model(cfg, 11)
    for n in (0, 1, 2):
        exp = ref.hidden(batch["input_ids"], batch["input_mask"], None, n).numpy()
        got = model.hidden_states(batch["input_ids"], batch["input_mask"], None, n).cpu().numpy()
        valid = batch["input_mask"].numpy().astype(bool)
        _close(got[valid], exp[valid])


def test_load_saved_and_module_prefix(tmp_path):
    from otter_amd import encoder as E
    cfg, seed, batch, expected, _ = U.load_golden("rob2_first")
    sd = U.state_dict_for(cfg, seed)
    ck = tmp_path / "checkpoint_best.pt"
    torch.save({"module." + k: v for k, v in sd.items()}, ck)
    model = E.RobertaSingleEncoderThreeCatPool(U.hf_like_config(cfg), U.ref_args(cfg))
    model = E.load_saved(model, str(ck), exact=False)
    got = model(E.move_to_cuda(batch))["embed"].cpu().numpy()
    _close(got, expected)


def test_ctx_encoder_surface():
    from types import SimpleNamespace
    from otter_amd import encoder as E
    cfg, _, _, _, _ = U.load_golden("bert2_ctx")
    with pytest.raises(NotImplementedError):
        E.CtxEncoder(U.hf_like_config(cfg), SimpleNamespace(multi_vector=3, scheme="tokenwise"))
    m = E.CtxEncoder(U.hf_like_config(cfg), SimpleNamespace(multi_vector=1, scheme="none"))
    assert all(k.startswith(("encoder_c.", "project.")) for k in m.state_dict_keys())
    r = E.RobertaCtxEncoder(U.hf_like_config(cfg), SimpleNamespace(no_proj=True))
    assert r.no_proj is False          # the projection head of RobertaCtxEncoder is not optional


def test_errors_are_loud():
    from otter_amd import encoder as E
    cfg, seed, batch, _, _ = U.load_golden("rob2_first")
    model = E.RobertaSingleEncoderThreeCatPool(U.hf_like_config(cfg), U.ref_args(cfg))
    with pytest.raises(Exception):
        model({k: v.cuda() for k, v in batch.items()})          # no weights loaded
    sd = dict(U.state_dict_for(cfg, seed))
    sd["encoder.embeddings.word_embeddings.weight"] = sd["encoder.embeddings.word_embeddings.weight"][:10]
    model.load_state_dict(sd, strict=False)
    with pytest.raises(ValueError):
        model({k: v.cuda() for k, v in batch.items()})          # wrong shape is refused by the library


@pytest.mark.parametrize("B,L", [(3, 200), (2, 512), (5, 70), (1, 64), (2, 333)])
def test_bf16_hidden_states_track_fp32_path(B, L):
    # every valid row of the bf16 path's hidden state against the fp32 path of the same library
    # (itself pinned to the reference fixtures above), after 1 and 2 layers, ragged batch:
    # bf16 rounding gives ~1e-2 absolute error on O(1) activations; a wrong key, mask or tile
    # shows as O(1).
    cfg = R.make_config("roberta", layers=2, vocab_size=1000)
    batch = R.make_batch(cfg, B, L, 31 + L)
    m32 = _model(cfg, 11)
    m16 = _model(cfg, 11, dtype="bf16")
    valid = batch["input_mask"].numpy().astype(bool)
    for n in (1, 2):
        ref = m32.hidden_states(batch["input_ids"], batch["input_mask"], None, n).cpu().numpy()[valid]
        got = m16.hidden_states(batch["input_ids"], batch["input_mask"], None, n).cpu().numpy()[valid]
        err = np.abs(got - ref)
        assert np.isfinite(got).all()
        assert err.max() < 0.15 and err.mean() < 0.012, (n, float(err.max()), float(err.mean()))


@pytest.mark.parametrize("family,pooling,L", [("roberta", "first", 512), ("roberta", "mean", 256), ("bert", "first", 128),
                                              ("roberta", "cls1", 384)])
def test_packed_batches_equal_the_dense_layout(family, pooling, L):
    # bf16 path, ragged batch (one full-length row, one 1-token row, one row whose part3 mask is empty and one
    # whose part mask reaches past input_mask): the packed layout skips the padded rows, the dense layout
    # (OTTER_NO_PACK=1) computes them like the reference; the embeddings must agree to fp32 rounding of the
    # pooling (every per-row operation is the same instruction sequence on the same data in both layouts)
    import os
    three = pooling != "cls1"
    cfg = R.make_config(family, layers=2, vocab_size=2000, three_cat=three, part_pooling=pooling if three else "first")
    B = 6
    batch = R.make_batch(cfg, B, L, 5 + L)
    lens = [L, 1, 70, 64, L - 1, 130]
    for b, n in enumerate(lens):
        batch["input_mask"][b, :] = 0
        batch["input_mask"][b, :n] = 1
        batch["input_ids"][b, n:] = cfg["pad_token_id"]
        for k in ("part2_mask", "part3_mask"):
            if k in batch:
                batch[k][b, n:] = 0
    if three:
        batch["part3_mask"][2, :] = 0                   # empty part: 'first' falls back to row 0, 'mean' gives NaN
        batch["part2_mask"][3, 100] = 1                 # a part position past the last input_mask position
    dev = {k: v.cuda() for k, v in batch.items()}
    model = _model(cfg, 3, dtype="bf16")
    packed = model(dev)["embed"].cpu().numpy()
    os.environ["OTTER_NO_PACK"] = "1"
    try:
        dense = model(dev)["embed"].cpu().numpy()
    finally:
        del os.environ["OTTER_NO_PACK"]
    assert packed.shape == dense.shape
    assert (np.isnan(packed) == np.isnan(dense)).all()
    ok = ~np.isnan(dense)
    assert ok.any()
    np.testing.assert_allclose(packed[ok], dense[ok], rtol=1e-5, atol=1e-5)
    # and a second, shorter batch through the same (grown) workspace
    dev2 = {k: v[:2].contiguous() for k, v in dev.items()}
    again = model(dev2)["embed"].cpu().numpy()
    ok2 = ~np.isnan(dense[:2])
    np.testing.assert_allclose(again[ok2], dense[:2][ok2], rtol=1e-5, atol=1e-5)


def _hidden_bf16(m16, batch, n, safe):
    import os
    if safe:
        os.environ["OTTER_ATTN_SAFE"] = "1"
    try:
        return m16.hidden_states(batch["input_ids"], batch["input_mask"], None, n).cpu().numpy()
    finally:
        os.environ.pop("OTTER_ATTN_SAFE", None)


@pytest.mark.parametrize("L", [64, 128, 320, 512])
def test_bf16_attention_pipeline_against_plain_kernel(L):
    # the persistent, pipelined attention kernel (no running maximum; accepted by its row sums) against the plain
    # online-softmax kernel of the same library on the same bf16 operands: sequence lengths that leave waves and
    # blocks without queries, ragged key padding, several items per workgroup (B * heads > CUs is not needed: the
    # item loop runs whenever B * heads > grid)
    cfg = R.make_config("roberta", layers=1, vocab_size=1000)
    B = 5
    batch = R.make_batch(cfg, B, L, 77 + L)
    lens = [L, max(1, L // 3), L - 1, 33 if L > 33 else L, max(1, L - 65)]
    for b, n in enumerate(lens):
        batch["input_mask"][b, :] = 0
        batch["input_mask"][b, :n] = 1
    m16 = _model(cfg, 13, dtype="bf16")
    valid = batch["input_mask"].numpy().astype(bool)
    got = _hidden_bf16(m16, batch, 1, False)[valid]
    ref = _hidden_bf16(m16, batch, 1, True)[valid]
    assert np.isfinite(got).all()
    err = np.abs(got - ref)
    assert err.max() < 0.1 and err.mean() < 0.006, (float(err.max()), float(err.mean()))


def test_bf16_attention_rejected_blocks_are_recomputed():
    # scores far outside what 2^s can hold without a running maximum (query projection x 80: |s| of several hundred):
    # the pipelined kernel's row sums overflow, it flags the blocks and the plain kernel recomputes them -- the result
    # must be what the plain kernel alone gives; and a sequence with every key masked must come out finite
    cfg = R.make_config("roberta", layers=1, vocab_size=1000)
    B, L = 3, 256
    batch = R.make_batch(cfg, B, L, 5)
    batch["input_mask"][:, :] = 1
    batch["input_mask"][1, :] = 0                      # every key masked
    batch["input_mask"][2, 100:] = 0
    sd = dict(U.state_dict_for(cfg, 17))
    for k in list(sd):
        if "attention.self.query" in k:
            sd[k] = sd[k] * 80.0
    from otter_amd import encoder as E
    m16 = E.RobertaSingleEncoder(U.hf_like_config(cfg), U.ref_args(cfg), dtype="bf16")
    m16.load_state_dict(sd, strict=False)
    m16.to("cuda:0").eval()
    got = _hidden_bf16(m16, batch, 1, False)
    ref = _hidden_bf16(m16, batch, 1, True)
    assert np.isfinite(got).all() and np.isfinite(ref).all()
    rows = [0, 2]   # (blocks whose sums stayed in range keep the pipelined kernel's result: bf16 rounding apart)
    err = np.abs(got[rows] - ref[rows])
    assert err.max() < 0.05 and (err == 0).mean() > 0.5, (float(err.max()), float((err == 0).mean()))
