#!/usr/bin/env python3
"""oracle/gen_host_goldens_meta.py — the 768-d (one-vector) dataset of the reference, run on the inputs that
tests/golden/host_features.json already holds:

    tests/golden/host_features_meta.json   retrieval/data/tr_dataset.py:58-81 (convert_tb_to_features_bert_metadata)
                                           driven through retrieval/data/encode_datasets.py:191-226 (EmDatasetMeta)
                                           + em_collate_bert :393-417

Same rules as oracle/gen_host_goldens.py (build container only; the reference is imported, never copied; the
unused `pathos` import is stubbed outside the reference tree).  The committed toy tokenizer is loaded, not
retrained, so the existing fixtures stay as they are.  TEST INFRASTRUCTURE ONLY.

    python oracle/gen_host_goldens_meta.py
"""
import json
import os
import sys
import tempfile
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
GOLD = os.path.join(ROOT, "tests", "golden")


def main():
    import pandas as pd
    from transformers import PreTrainedTokenizerFast
    tmp = tempfile.mkdtemp(prefix="otter_hostgold_")
    os.makedirs(os.path.join(tmp, "pathos"))
    open(os.path.join(tmp, "pathos", "__init__.py"), "w").close()
    with open(os.path.join(tmp, "pathos", "multiprocessing.py"), "w") as f:
        f.write("from multiprocessing import Pool\n")
    sys.path.insert(0, tmp)
    sys.path.insert(0, REF)
    from retrieval.data import encode_datasets as ref_ds          # noqa: E402  (the reference)

    class Tok45(PreTrainedTokenizerFast):
        """transformers 4.5 API the reference calls (tr_dataset.py:66): encode_plus == __call__ in 5.x"""
        def encode_plus(self, text, text_pair=None, **kw):
            return self(text, text_pair=text_pair, **kw)

    tokenizer = Tok45.from_pretrained(os.path.join(GOLD, "toy_tokenizer"))
    with open(os.path.join(GOLD, "host_features.json")) as f:
        feats = json.load(f)
    corpus, questions = feats["corpus"], feats["questions"]
    cases = []
    for max_c_len, max_q_len, normalize_table in ((128, 24, True), (64, 16, False), (512, 70, True)):
        args = SimpleNamespace(model_name="roberta-toy", max_c_len=max_c_len, max_q_len=max_q_len,
                               normalize_table=normalize_table, no_passages=False, psg_mode="ori", embed_save_path=tmp)
        for table in (True, False):
            ds = ref_ds.EmDatasetMeta(tokenizer, "unused", table, args)
            if table:
                ds.data = [dict(c, table=pd.DataFrame(data=c["table"][1], columns=c["table"][0])) for c in corpus]
            else:
                ds.data = questions
            items = [ds[i] for i in range(len(ds.data))]
            batch = ref_ds.em_collate_bert(items, pad_id=tokenizer.pad_token_id)
            cases.append({"max_c_len": max_c_len, "max_q_len": max_q_len, "normalize_table": normalize_table,
                          "table": table, "batch": {k: v.tolist() for k, v in batch.items()}})
    out = os.path.join(GOLD, "host_features_meta.json")
    with open(out, "w") as f:
        json.dump({"inputs": "tests/golden/host_features.json: corpus, questions", "cases": cases}, f)
    print("wrote", out, [sorted(c["batch"]) for c in cases[:2]])


if __name__ == "__main__":
    main()
