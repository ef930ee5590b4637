#!/usr/bin/env python3
"""oracle/gen_host_goldens.py — run the REFERENCE's host-side functions (feature builder, mask rule,
string metrics) on synthetic inputs and commit inputs + outputs as fixtures:

    tests/golden/host_features.json   retrieval/data/tr_dataset.py:29-46,117-153 driven through
                                      retrieval/data/encode_datasets.py:228-256 + em_collate_bert :393-417
    tests/golden/host_metrics.json    retrieval/utils/utils.py:135-153 (para_has_answer), :261-268 (found_table)
                                      with retrieval/drqa/drqa_tokenizers/simple_tokenizer.py
    tests/golden/toy_tokenizer/       a byte-level BPE trained HERE on the synthetic texts (no pretrained
                                      vocabulary is available offline), RoBERTa-style special tokens

Runs only in the build container (needs /root/reference; the reference cannot travel).  The one module
the reference imports but does not ship (pathos, retrieval/data/tr_dataset.py:23, unused) is stubbed on
a temporary sys.path entry outside the reference tree.  TEST INFRASTRUCTURE ONLY.

    python oracle/gen_host_goldens.py
"""
import json
import os
import sys
import tempfile
from types import SimpleNamespace

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
GOLD = os.path.join(ROOT, "tests", "golden")


def synth_corpus(n, seed):
    rs = np.random.RandomState(seed)
    words = ("alpha beta gamma delta river mountain league season 1998 2004 team coach city Paris "
             "Tokyo étoile naïve São-Paulo record final winner score points club year born film "
             "director album song label U.S. 3-2 (draw) “quoted” co-operative").split()

    def sent(k):
        return " ".join(words[i] for i in rs.randint(0, len(words), size=k))

    corpus = []
    for i in range(n):
        ncol = int(rs.randint(2, 6))
        header = [sent(int(rs.randint(1, 3))) for _ in range(ncol)]
        row = [sent(int(rs.randint(1, 4))) for _ in range(ncol)]
        npsg = int(rs.randint(0, 11))                       # up to 10: the dataset keeps 8
        passages = [sent(int(rs.randint(5, 60))) + "." for _ in range(npsg)]
        corpus.append({"table_id": "Table_%d_%d" % (i // 3, i % 3), "row_id": i % 3,
                       "table": [header, [row]], "passages": passages,
                       "passages_id": ["/wiki/P%d" % j for j in range(npsg)],
                       "meta_data": {"title": sent(3), "section_title": sent(2), "header": header},
                       "url": "https://example.org/%d" % i, "title": sent(3), "intro": sent(12)})
    questions = [{"question_id": "q%d" % i, "question": sent(int(rs.randint(4, 40))) + ("?" if i % 2 == 0 else ""),
                  "table_id": "Table_%d_0" % i, "answer-text": sent(2)} for i in range(8)]
    return corpus, questions


def train_toy_tokenizer(texts, out_dir):
    from tokenizers import Tokenizer, decoders, models, pre_tokenizers, processors, trainers
    from transformers import PreTrainedTokenizerFast
    tok = Tokenizer(models.BPE(unk_token=None))
    tok.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False)
    tok.decoder = decoders.ByteLevel()
    trainer = trainers.BpeTrainer(vocab_size=400, special_tokens=["<s>", "<pad>", "</s>", "<unk>", "<mask>"],
                                  initial_alphabet=pre_tokenizers.ByteLevel.alphabet(), show_progress=False)
    tok.train_from_iterator(texts, trainer)
    tok.post_processor = processors.TemplateProcessing(
        single="<s> $A </s>", pair="<s> $A </s> </s> $B </s>",
        special_tokens=[("<s>", tok.token_to_id("<s>")), ("</s>", tok.token_to_id("</s>"))])
    class Tok45(PreTrainedTokenizerFast):
        """transformers 4.5 API the reference calls (tr_dataset.py:139): encode_plus == __call__ in 5.x"""
        def encode_plus(self, text, text_pair=None, **kw):
            return self(text, text_pair=text_pair, **kw)

    fast = Tok45(tokenizer_object=tok, bos_token="<s>", eos_token="</s>", cls_token="<s>",
                                   sep_token="</s>", pad_token="<pad>", unk_token="<unk>", mask_token="<mask>")
    fast.save_pretrained(out_dir)
    return fast


def main():
    import pandas as pd
    tmp = tempfile.mkdtemp(prefix="otter_hostgold_")
    os.makedirs(os.path.join(tmp, "pathos"))
    open(os.path.join(tmp, "pathos", "__init__.py"), "w").close()
    with open(os.path.join(tmp, "pathos", "multiprocessing.py"), "w") as f:
        f.write("from multiprocessing import Pool\n")
    sys.path.insert(0, tmp)
    sys.path.insert(0, REF)
    from retrieval.data import encode_datasets as ref_ds          # noqa: E402  (the reference)
    from retrieval.data import tr_dataset as ref_tr               # noqa: E402
    # the package's __init__ imports a CoreNLP wrapper that needs `pexpect` (absent here): load the two
    # files para_has_answer's tokenizer lives in under a bare package object instead
    import importlib
    import types
    pkg = types.ModuleType("ref_drqa_tokenizers")
    pkg.__path__ = [os.path.join(REF, "retrieval", "drqa", "drqa_tokenizers")]
    sys.modules["ref_drqa_tokenizers"] = pkg
    RefTok = importlib.import_module("ref_drqa_tokenizers.simple_tokenizer").SimpleTokenizer
    from retrieval.utils import utils as ref_utils                # noqa: E402

    corpus, questions = synth_corpus(14, 5)
    texts = []
    for c in corpus:
        texts += c["passages"] + c["table"][0] + c["table"][1][0] + [c["meta_data"]["title"]]
    texts += [q["question"] for q in questions] + ["[TAB] [TITLE] [SECTITLE] [DATA] [PASSAGE] [SEP] [HEADER] is ."]
    tok_dir = os.path.join(GOLD, "toy_tokenizer")
    tokenizer = train_toy_tokenizer(texts, tok_dir)

    cases = []
    for max_c_len, max_q_len, normalize_table in ((128, 24, True), (64, 16, False), (512, 70, True)):
        args = SimpleNamespace(model_name="roberta-toy", max_c_len=max_c_len, max_q_len=max_q_len,
                               normalize_table=normalize_table, one_query=True, no_passages=False, psg_mode="ori",
                               embed_save_path=tmp)
        for table in (True, False):
            ds = ref_ds.EmDatasetMetaThreeCat(tokenizer, "unused", table, args)
            if table:
                ds.data = [dict(c, table=pd.DataFrame(data=c["table"][1], columns=c["table"][0])) for c in corpus]
            else:
                ds.data = questions
            items = [ds[i] for i in range(len(ds.data))]
            batch = ref_ds.em_collate_bert(items, pad_id=tokenizer.pad_token_id)
            strings = None
            if table:
                fn = ref_tr.convert_tb_to_string_metadata_norm if normalize_table else ref_tr.convert_tb_to_string_metadata
                strings = [list(fn(d["table"], d["passages"][:8], d["meta_data"])) for d in ds.data]
            cases.append({"max_c_len": max_c_len, "max_q_len": max_q_len, "normalize_table": normalize_table,
                          "table": table, "strings": strings,
                          "batch": {k: v.tolist() for k, v in batch.items()}})
    with open(os.path.join(GOLD, "host_features.json"), "w") as f:
        json.dump({"corpus": corpus, "questions": questions, "cases": cases}, f)

    # string metrics
    rtok = RefTok()
    blocks = [{"table_id": c["table_id"], "passages": c["passages"], "table": c["table"]} for c in corpus]
    mcases = []
    rs = np.random.RandomState(9)
    answers = ["Paris", "paris", "São-Paulo", "sao-paulo", "U.S.", "3-2", "( draw )", "“quoted”", "no such answer",
               "étoile naïve", "1998", "co-operative", "league season", ""]
    for c in corpus[:6]:
        w = (" ".join(c["passages"]) or "x y").split()
        s = int(rs.randint(0, max(1, len(w) - 2)))
        answers.append(" ".join(w[s:s + 2]))
        answers.append(c["table"][1][0][0])
    for a in answers:
        for lo, hi in ((0, 14), (0, 3), (5, 9)):
            tbs = blocks[lo:hi]
            mcases.append({"answer": a, "lo": lo, "hi": hi,
                           "para_has_answer": bool(ref_utils.para_has_answer(a, tbs, rtok))})
    tcases = []
    for gold in ("Table_0_0", "Table_0", "Table_3_1,Table_99_0", "Table_99_0", "able_2", "Table_4_2,", ""):
        for lo, hi in ((0, 14), (0, 3), (9, 12)):
            tcases.append({"gold": gold, "lo": lo, "hi": hi, "found_table": bool(ref_utils.found_table(gold, blocks[lo:hi]))})
    with open(os.path.join(GOLD, "host_metrics.json"), "w") as f:
        json.dump({"blocks": blocks, "para_has_answer": mcases, "found_table": tcases,
                   "words": [{"text": t, "words": rtok.tokenize(ref_utils.normalize(t)).words(uncased=True)}
                             for t in answers + [c["passages"][0] for c in corpus if c["passages"]][:4]]}, f)
    print("wrote", os.path.join(GOLD, "host_features.json"), os.path.join(GOLD, "host_metrics.json"), tok_dir)


if __name__ == "__main__":
    main()
