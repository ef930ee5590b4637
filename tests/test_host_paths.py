"""CPU tests of the host-side "next" rows against fixtures produced by the REFERENCE's
own functions (oracle/gen_host_goldens.py -> tests/golden/host_*.json, tests/golden/toy_tokenizer/):
feature builder + mask rule + collate, string metrics, recall arithmetic, id2doc / embedding formats."""
import json
import os

import numpy as np
import pytest
import torch

from otter_amd import eval_driver as EV
from otter_amd import features as F
from otter_amd.encode_driver import em_collate_bert

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def feats():
    with open(os.path.join(GOLD, "host_features.json")) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def metrics():
    with open(os.path.join(GOLD, "host_metrics.json")) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def tokenizer():
    from transformers import PreTrainedTokenizerFast
    return PreTrainedTokenizerFast.from_pretrained(os.path.join(GOLD, "toy_tokenizer"))


def test_feature_builder_and_collate_match_reference(feats, tokenizer):
    from types import SimpleNamespace
    for case in feats["cases"]:
        args = SimpleNamespace(max_c_len=case["max_c_len"], max_q_len=case["max_q_len"],
                               normalize_table=case["normalize_table"], one_query=True, no_passages=False,
                               psg_mode="ori")
        data = feats["corpus"] if case["table"] else feats["questions"]
        ds = F.EmDatasetMetaThreeCat(tokenizer, data, case["table"], args)
        batch = em_collate_bert([ds[i] for i in range(len(ds))], pad_id=tokenizer.pad_token_id)
        assert set(batch) == set(case["batch"])
        for k, v in case["batch"].items():
            exp = torch.tensor(v)
            assert batch[k].dtype == torch.int64 and tuple(batch[k].shape) == tuple(exp.shape), k
            assert torch.equal(batch[k], exp), (k, case["max_c_len"], case["table"])
        if case["strings"] is not None:
            fn = F.convert_tb_to_string_metadata_norm if case["normalize_table"] else F.convert_tb_to_string_metadata
            got = [list(fn(d["table"], d["passages"][:8], d["meta_data"])) for d in data]
            assert got == case["strings"]


def test_one_vector_dataset_matches_reference(feats, tokenizer):
    """EmDatasetMeta (the 768-d encoders' dataset) against the reference's own, same inputs
    (oracle/gen_host_goldens_meta.py -> tests/golden/host_features_meta.json)."""
    from types import SimpleNamespace
    with open(os.path.join(GOLD, "host_features_meta.json")) as f:
        meta = json.load(f)
    assert len(meta["cases"]) == 6
    for case in meta["cases"]:
        args = SimpleNamespace(max_c_len=case["max_c_len"], max_q_len=case["max_q_len"],
                               normalize_table=case["normalize_table"], no_passages=False, psg_mode="ori")
        data = feats["corpus"] if case["table"] else feats["questions"]
        ds = F.EmDatasetMeta(tokenizer, data, case["table"], args)
        batch = em_collate_bert([ds[i] for i in range(len(ds))], pad_id=tokenizer.pad_token_id)
        assert set(batch) == set(case["batch"]) == {"input_ids", "input_mask"}
        for k, v in case["batch"].items():
            assert torch.equal(batch[k], torch.tensor(v)), (k, case["max_c_len"], case["table"])


def test_mask_rule_on_synthetic_special_token_masks():
    # <s> A A A </s> </s> B B </s> <pad> <pad>: part2 = tokens of A, part3 = tokens of B
    stm = [1, 0, 0, 0, 1, 1, 0, 0, 1, 1, 1]
    p2, p3 = F.threecat_masks(stm, question=False)
    assert p2.tolist() == [0, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0]
    assert p3.tolist() == [0, 0, 0, 0, 0, 0, 1, 1, 0, 0, 0]
    # question: <s> Q Q </s> <pad>: part3 == part2
    p2, p3 = F.threecat_masks([1, 0, 0, 1, 1], question=True)
    assert p2.tolist() == [0, 1, 1, 0, 0] and p3.tolist() == p2.tolist()
    # empty second segment: <s> A </s> </s> </s>: part3 is empty (the encoder's 'first' pooling then takes row 0)
    p2, p3 = F.threecat_masks([1, 0, 1, 1, 1], question=False)
    assert p2.tolist() == [0, 1, 0, 0, 0] and p3.sum() == 0


def test_string_metrics_match_reference(metrics):
    tok = EV.SimpleTokenizer()
    blocks = metrics["blocks"]
    for w in metrics["words"]:
        assert tok.words(EV.normalize(w["text"]), uncased=True) == w["words"]
    for c in metrics["para_has_answer"]:
        assert EV.para_has_answer(c["answer"], blocks[c["lo"]:c["hi"]], tok) == c["para_has_answer"], c["answer"]
    for c in metrics["found_table"]:
        assert EV.found_table(c["gold"], blocks[c["lo"]:c["hi"]]) == c["found_table"], c["gold"]
    assert any(c["para_has_answer"] for c in metrics["para_has_answer"])
    assert any(not c["para_has_answer"] for c in metrics["para_has_answer"])


def test_records_recall_and_jsonl(tmp_path, metrics):
    blocks = metrics["blocks"]
    id2doc = {str(i): b for i, b in enumerate(blocks)}
    ds_items = [{"question_id": "q%d" % i, "question": "what %d?" % i, "table_id": blocks[i]["table_id"],
                 "answer-text": blocks[i]["table"][1][0][1], "type": "single" if i % 2 else "multi"} for i in range(4)]
    beam = 5
    I = np.array([[0, 1, 2, 3, 4], [4, 1, 0, 2, 3], [5, 6, 7, 8, 9], [9, 8, 3, 6, 5]], np.int64)
    m, o = EV.records(ds_items, I, id2doc, beam, eval_only_ans=True, with_outputs=True)
    assert [x["table_recall"] for x in m] == [1, 1, 0, 1]
    # (the reference joins passages, header and row WITHOUT separators, retrieval/utils/utils.py:143-145, so a
    # first-cell answer fuses with the last header word; a second-cell answer is found)
    assert [x["ans_recall"] for x in m] == [1, 1, int(EV.para_has_answer(ds_items[2]["answer-text"], blocks[5:10],
                                                                          EV.SimpleTokenizer())), 1]
    assert list(o[0].keys()) == ["question_id", "question", "top_5", "answer-text", "table_id"]
    assert [t["table_id"] for t in o[1]["top_5"]] == [blocks[j]["table_id"] for j in I[1]]
    rep = EV.table_recall_report(o, "top_5")
    # labels as the reference prints them: n=3 is reported as "@5", n=25 as "@50"
    assert [r[0] for r in rep] == ["1", "5", "10", "15", "20", "50", "100"]
    by = {r[0]: r for r in rep}
    assert by["1"][1:] == (0.25, 1, 4)            # only question 0 has its table first
    assert by["5"][1:] == (0.75, 3, 4)            # within the first THREE: questions 0, 1 (row 1), 3 (row 2)
    assert by["100"][1:] == (0.75, 3, 4)
    a = EV.answer_recall_report(m)
    assert a["table_recall"] == 0.75 and set(a["types"]) == {"single", "multi"}
    p = tmp_path / "out.jsonl"
    EV.write_jsonl(str(p), o)
    back = [json.loads(line) for line in open(p)]
    assert back == o
    # legacy id2doc ([title, text] lists) is converted as the reference does
    lp = tmp_path / "id2doc.json"
    json.dump({"0": ["T", "x"], "1": ["U", "y"]}, open(lp, "w"))
    assert EV.load_id2doc(str(lp))["1"] == {"title": "U", "text": "y"}


def test_id2doc_and_embedding_formats(tmp_path, feats):
    corpus = feats["corpus"]
    path = tmp_path / "emb" / "id2doc.json"
    F.write_id2doc(str(path), corpus)
    id2doc = json.load(open(path))
    assert len(id2doc) == len(corpus)
    d3 = id2doc["3"]
    assert list(d3.keys()) == ["table_id", "row_id", "table", "meta_data", "passages_id", "passages"]
    assert d3["table"] == corpus[3]["table"] and d3["passages"] == corpus[3]["passages"]
    # embeddings: reference .npy and the row-sharded store written by the data-parallel encode
    rng = np.random.RandomState(0)
    full = rng.randn(10, 6).astype(np.float32)
    np.save(tmp_path / "all.npy", full)
    assert np.array_equal(np.asarray(F.open_embeddings(str(tmp_path / "all.npy"))), full)
    bounds = [(0, 4), (4, 8), (8, 10)]
    for r, (lo, hi) in enumerate(bounds):
        F.write_embedding_shards(str(tmp_path / "sh" / "corpus"), torch.from_numpy(full[lo:hi]), lo, 3, r)
    shards = F.open_embedding_shards(str(tmp_path / "sh" / "corpus"))
    assert [s[0] for s in shards] == [0, 4, 8]
    assert np.array_equal(np.concatenate([np.asarray(s[1]) for s in shards]), full)
    os.remove(tmp_path / "sh" / "corpus.shard00001.npy")
    os.remove(tmp_path / "sh" / "corpus.shard00001.json")
    with pytest.raises(ValueError):
        F.open_embedding_shards(str(tmp_path / "sh" / "corpus"))


def test_whitening_matches_the_reference_golden():
    """otter_amd.whitening against outputs of the reference's own functions (oracle/gen_whitening_golden.py).
    The SVD fixes the kernel's columns only up to sign, so the comparison is on what the search sees: the
    Gram matrix q' x'^T (invariant to those signs), plus the defining property (unit covariance)."""
    import numpy as np
    from otter_amd import whitening as W
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "whitening.npz"))
    w1, w2 = W.whitening(g["e1"], g["e2"])
    assert w1.dtype == np.float32 and w2.dtype == np.float32 and w1.shape == g["w1"].shape
    np.testing.assert_allclose(w2 @ w1.T, g["w2"] @ g["w1"].T, rtol=2e-3, atol=2e-3)
    kernel, bias = W.compute_kernel_bias(np.concatenate([g["e1"], g["e2"]], 0))
    np.testing.assert_allclose(bias, g["bias"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(kernel @ kernel.T, g["kernel"] @ g["kernel"].T, rtol=1e-6, atol=1e-8)
    allv = np.concatenate([w1, w2], 0).astype(np.float64)
    np.testing.assert_allclose(np.cov(allv.T), np.eye(allv.shape[1]), atol=5e-3)
    assert W.transform_and_normalize(g["e1"]) is g["e1"] or np.array_equal(W.transform_and_normalize(g["e1"]), g["e1"])


def test_index_file_round_trip_and_validation(tmp_path):
    """otter_amd.index_io on the host: faiss' IndexFlatIP file layout as restated there (parity unpinned against
    real faiss files: none ship with the reference and faiss is absent), both vector-count conventions, and
    loud refusal of truncated / foreign files."""
    import struct
    from otter_amd import index_io as IO

    class Fake:
        def __init__(self, x):
            self.x, self.d, self.ntotal = x, x.shape[1], x.shape[0]

        def reconstruct_n(self, i0, n):
            return self.x[i0:i0 + n]

    x = np.random.RandomState(4).randn(1000, 24).astype(np.float32)
    p = str(tmp_path / "flat.index")
    IO.write_index(Fake(x), p)
    d, xb = IO.read_flat(p)
    assert d == 24 and xb.shape == (1000, 24) and np.array_equal(np.asarray(xb), x)
    raw = bytearray(open(p, "rb").read())
    assert raw[:4] == b"IxFI" and len(raw) == 37 + 8 + x.nbytes
    # faiss <= 1.7.0 stores the count in units of 4 floats
    old = bytearray(raw)
    old[37:45] = struct.pack("<Q", x.size // 4)
    open(p, "wb").write(old)
    assert np.array_equal(np.asarray(IO.read_flat(p)[1]), x)
    for bad in (raw[:-4], b"IHNf" + bytes(raw[4:]), bytes(raw[:37]) + struct.pack("<Q", 7) + bytes(raw[45:])):
        open(p, "wb").write(bad)
        with pytest.raises(ValueError):
            IO.read_flat(p)
    IO.write_index(Fake(x[:0]), p)
    assert IO.read_flat(p)[1].shape == (0, 24)


def test_encode_script_arguments_and_class_choice(tmp_path):
    """encode_driver: argument names / defaults of retrieval/config.py:encode_args, the class choice of
    scripts/encode_corpus.py:48-69, and the record readers (no GPU involved)."""
    from otter_amd import encode_driver as ED
    from otter_amd import encoder as E
    a = ED.encode_args(["--predict_file", "p", "--init_checkpoint", "c", "--embed_save_path", "e"])
    assert (a.max_c_len, a.max_q_len, a.predict_batch_size, a.part_pooling, a.psg_mode) == (512, 70, 512, "first", "ori")
    assert a.three_cat and a.no_proj and a.normalize_table and a.metadata and not a.encode_table and not a.shared_encoder
    cfg = dict(vocab_size=64, hidden_size=64, num_hidden_layers=1, num_attention_heads=1, intermediate_size=128,
               max_position_embeddings=66, type_vocab_size=1, pad_token_id=1, layer_norm_eps=1e-5)
    from types import SimpleNamespace
    hf = SimpleNamespace(**cfg)
    for name, extra, cls in (("x/roberta-base", [], E.RobertaSingleEncoderThreeCatPool),
                             ("bert-base-uncased", [], E.SingleEncoderThreeCatPool),
                             ("roberta-base", ["--part_pooling", "att_self_pool"], E.RobertaSingleEncoderThreeCatAtt),
                             ("roberta-base", ["--one_vector"], E.RobertaSingleEncoder),
                             ("bert-base-uncased", ["--one_vector"], E.SingleEncoder)):
        a = ED.encode_args(["--predict_file", "p", "--init_checkpoint", "c", "--embed_save_path", "e", "--model_name", name] + extra)
        assert type(ED.build_model(hf, a)) is cls
    recs = [{"question": "a b?"}, {"question": "c"}]
    with open(tmp_path / "q.json", "w") as f:
        json.dump(recs, f)
    with open(tmp_path / "q.jsonl", "w") as f:
        f.write("\n".join(json.dumps(r) for r in recs) + "\n\n")
    assert ED.load_records(str(tmp_path / "q.json"), False) == recs == ED.load_records(str(tmp_path / "q.jsonl"), False)
    with pytest.raises(ValueError):
        ED.load_records(str(tmp_path / "corpus.pkl"), True)


def test_tfidf_subsets_follow_the_reference_mapping(feats):
    """eval_driver.tfidf_subsets = `mapping` of retrieval/eval_ottqa_retrieval_tfidffilter.py:486-493: the rows whose
    table_id is among the first topk_doc TF-IDF documents of the question, in id2doc order."""
    from otter_amd import eval_driver as EV
    corpus, questions = feats["corpus"], feats["questions"]
    id2doc = {str(k): v for k, v in F.id2doc_from_corpus(corpus).items()}
    tables = sorted({c["table_id"] for c in corpus})
    tfidf = {qd["question_id"]: {"doc_ids": [tables[(2 * i + j) % len(tables)] for j in range(5)] + ["absent"]}
             for i, qd in enumerate(questions)}
    for topk in (1, 3, 6):
        subs = EV.tfidf_subsets(questions, id2doc, tfidf, topk)
        assert len(subs) == len(questions)
        for qd, sub in zip(questions, subs):
            docs = set(tfidf[qd["question_id"]]["doc_ids"][:topk])
            within = [int(idx) for idx, doc in id2doc.items() if doc["table_id"] in docs]     # the reference's scan
            assert sub.dtype == np.int64 and sub.tolist() == within
