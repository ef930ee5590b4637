"""CPU: the restatement oracle/encoder_ref.py must reproduce the fixtures the REFERENCE classes
produced (tests/golden/encoder_*.npz, see oracle/gen_encoder_goldens.py).  Tolerance: 2e-5
absolute on embeddings of magnitude <= ~5 (fp32, different BLAS blocking / thread counts)."""
import numpy as np
import pytest

from oracle import encoder_ref as R
from tests import _enc_util as U


@pytest.mark.parametrize("name", U.golden_names())
def test_restatement_matches_reference_fixture(name):
    cfg, seed, batch, expected, extra = U.load_golden(name)
    klass, q_seed, encode_table = U.golden_meta(name)
    ref = R.EncoderRef(cfg, U.weights_for(cfg, seed, q_seed, encode_table)[1])
    got = ref(batch)["embed"].numpy()
    assert got.shape == expected.shape
    scale = max(1.0, float(np.abs(expected).max()))
    assert np.abs(got - expected).max() <= 2e-5 * scale
    if extra:
        for li, l in enumerate(extra["probe_layers"]):
            h = ref.hidden(batch["input_ids"], batch["input_mask"], batch.get("input_type_ids"), int(l))
            h = h[:, extra["probe_pos"], :].numpy()
            assert np.abs(h - extra["probe_hidden"][li]).max() <= 2e-5 * max(1.0, np.abs(h).max())


def test_roberta_position_ids_follow_the_ids_not_the_mask():
    cfg = R.make_config("roberta", layers=1, vocab_size=50)
    sd = R.make_state_dict(cfg, 1)
    ref = R.EncoderRef(cfg, sd)
    import torch
    ids = torch.tensor([[0, 5, 1, 7, 2, 1]])      # a pad in the middle keeps position = pad_idx
    e = ref.embed(ids)
    ids2 = torch.tensor([[0, 5, 7, 2]])
    e2 = ref.embed(ids2)
    assert torch.allclose(e[0, [0, 1, 3, 4]], e2[0], atol=1e-6)


def test_first_pooling_empty_mask_and_b1():
    import torch
    h = torch.arange(2 * 4 * 3, dtype=torch.float32).view(2, 4, 3)
    m = torch.tensor([[0, 0, 1, 1], [0, 0, 0, 0]])
    out = R.pooling_masked_part(h, m, "first")
    assert torch.equal(out[0], h[0, 2]) and torch.equal(out[1], h[1, 0])
    assert R.pooling_masked_part(h[:1], m[:1], "first").shape == (1, 3)
