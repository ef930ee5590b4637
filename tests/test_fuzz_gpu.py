"""A fixed-seed slice of tools/fuzz_search.py: random corpus sizes, row lengths, batch sizes, k, plane settings and
score distributions (ties, duplicates, clusters), several batch sizes in a row on one index; D and I bit for bit
against oracle/flat_ip.c in the kernels' summation order.  The long runs (minutes, other seeds) are the tool's."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fuzz():
    spec = importlib.util.spec_from_file_location("fuzz_search", os.path.join(ROOT, "tools", "fuzz_search.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_random_shapes_match_the_oracle():
    assert _fuzz().main(["--seed", "3", "--cases", "60", "--seconds", "100", "--max-work", "1.5e10"]) == 0


def test_random_large_corpora_match_the_oracle():
    # corpora large enough for the coarse passes (bf16 planes, re-scoring, proof) and the optimistic start
    assert _fuzz().main(["--seed", "4", "--cases", "5", "--seconds", "100", "--min-n", "300000", "--max-work", "1e11"]) == 0


def test_random_encoder_shapes_match_the_restatement():
    # tools/fuzz_encoder.py: family, pooling, projection, class, batch, any sequence length <= 512, ragged rows; fp32
    # path at the fixtures' tolerance against oracle/encoder_ref.py, bf16 path by cosine
    spec = importlib.util.spec_from_file_location("fuzz_encoder", os.path.join(ROOT, "tools", "fuzz_encoder.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.main(["--seed", "2", "--cases", "80", "--seconds", "100"]) == 0
