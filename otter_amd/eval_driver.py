"""Retrieval evaluation driver: the counterpart of reference scripts/eval_ottqa_retrieval.py:56-203
around the HIP index.

What the reference does per question — `index.search(q[None], beam_size)` inside a 24-process
fork pool (:124-160) — becomes ONE batched `index.search(query_embeddings, beam_size)` in the
parent (a HIP context cannot be used after fork, and one batched call is what makes the scan
MFMA-bound); everything after the search is the reference's own CPU string work, restated:

    id2doc lookup                    scripts/eval_ottqa_retrieval.py:117-122,135-137
    metric_i / output_i records      :138-154   (JSONL schema {question_id, question, top_K, answer-text, table_id})
    Table Recall@n                   :170-192   (including its label quirks: n=3 is printed as "@5", n=25 as "@50")
    answer / table recall by type    :194-203
    found_table                      retrieval/utils/utils.py:261-268 (substring test of each gold id in tb['table_id'])
    para_has_answer                  retrieval/utils/utils.py:135-153 (token-sequence containment, NFD, uncased)
    SimpleTokenizer                  retrieval/drqa/drqa_tokenizers/simple_tokenizer.py:18-57 (the shipped equivalent of
                                     the missing retrieval/utils/basic_tokenizer.py the script imports at :20)

    python -m otter_amd.eval_driver --raw_data_path dev.json --query_embeddings_path q.npy \
        --corpus_embeddings_path c.npy --id2doc_path id2doc.json --output_save_path out.jsonl [--eval_only_ans]
"""
import argparse
import collections
import json
import os
import logging
import unicodedata

import numpy as np
import regex

logger = logging.getLogger("otter_amd.eval")

# (n, label) pairs exactly as the reference prints them (scripts/eval_ottqa_retrieval.py:179-192)
RECALL_POINTS = ((1, "1"), (3, "5"), (10, "10"), (15, "15"), (20, "20"), (25, "50"), (100, "100"))


def normalize(text):
    """retrieval/utils/utils.py:90-92."""
    return unicodedata.normalize("NFD", text)


class SimpleTokenizer:
    """retrieval/drqa/drqa_tokenizers/simple_tokenizer.py:18-57, reduced to what para_has_answer uses:
    tokenize(text).words(uncased=True)."""
    ALPHA_NUM = r"[\p{L}\p{N}\p{M}]+"
    NON_WS = r"[^\p{Z}\p{C}]"

    def __init__(self):
        self._regexp = regex.compile("(%s)|(%s)" % (self.ALPHA_NUM, self.NON_WS),
                                     flags=regex.IGNORECASE + regex.UNICODE + regex.MULTILINE)

    def words(self, text, uncased=False):
        toks = [m.group() for m in self._regexp.finditer(text)]
        return [t.lower() for t in toks] if uncased else toks


def found_table(answer, tbs):
    """retrieval/utils/utils.py:261-268: any comma-separated gold id that is a SUBSTRING of a block's table_id."""
    for g_tid in answer.split(","):
        for tb in tbs:
            if g_tid in tb["table_id"]:
                return True
    return False


def para_has_answer(answer_text, table_blocks, tokenizer):
    """retrieval/utils/utils.py:135-153."""
    answer = tokenizer.words(normalize(answer_text), uncased=True)
    for blk in table_blocks:
        text = " ".join(blk["passages"]) + " ".join(blk["table"][0]) + " ".join(blk["table"][1][0])
        words = tokenizer.words(normalize(text), uncased=True)
        for i in range(0, len(words) - len(answer) + 1):
            if answer == words[i: i + len(answer)]:
                return True
    return False


def load_id2doc(path):
    """scripts/eval_ottqa_retrieval.py:117-122."""
    with open(path) as f:
        id2doc = json.load(f)
    if isinstance(id2doc["0"], list):
        id2doc = {k: {"title": v[0], "text": v[1]} for k, v in id2doc.items()}
    return id2doc


def build_index(xb, d=None, devices=None):
    """scripts/eval_ottqa_retrieval.py:108-113: IndexFlatIP(d); add(xb) and, with --gpu,
    faiss.index_cpu_to_all_gpus(index) -- here the rows go straight to their GPUs (row shards, one process:
    index.MultiGpuIndexFlatIP).  xb may be a numpy memmap."""
    from .index import IndexFlatIP, MultiGpuIndexFlatIP
    d = xb.shape[1] if d is None else d
    if devices and len(devices) > 1:
        index = MultiGpuIndexFlatIP(d, devices=devices)
        index.add(xb)
        return index
    index = IndexFlatIP(d, device=0 if not devices else devices[0])
    step = 1 << 18                       # add in slabs: a memmapped .npy never has to fit in host memory
    for lo in range(0, xb.shape[0], step):
        index.add(np.ascontiguousarray(xb[lo:lo + step], dtype=np.float32))
    return index


def tfidf_subsets(ds_items, id2doc, tfidf_qid2docs, topk_doc=20):
    """retrieval/eval_ottqa_retrieval_tfidffilter.py:486-493 (`mapping`): per question, the rows whose block belongs to
    one of its first `topk_doc` TF-IDF documents, in id2doc order -- as row-id lists for IndexFlatIP.search_subsets.
    The reference scans the whole id2doc per question and copies the rows' embeddings into a dict it pickles; here
    one pass builds table_id -> rows and nothing is copied (the corpus is already resident in the index)."""
    by_table = collections.defaultdict(list)
    for idx in range(len(id2doc)):
        by_table[id2doc[str(idx)]["table_id"]].append(idx)
    subsets = []
    for item in ds_items:
        docs = set(tfidf_qid2docs[item["question_id"]]["doc_ids"][:topk_doc])
        subsets.append(np.array(sorted(r for t in docs for r in by_table.get(t, ())), dtype=np.int64))
    return subsets


def filtered_search(index, q, beam_size, subsets):
    """:530-546: the reference builds a throw-away IndexFlatIP over each question's rows and searches it with that one
    question; here ONE batched call on the resident corpus (otter_index_search_subset).  Returns I [nq, beam] of ROW
    ids (-1 padded when a question has fewer rows than beam): id2doc[str(row)] is the block the reference would
    have found at qid2tb[qid][str(position)]."""
    D, I = index.search_subsets(q, beam_size, subsets, return_positions=False)
    return D, I


def records(ds_items, I, id2doc, beam_size, eval_only_ans=False, with_outputs=True, tokenizer=None):
    """The per-question body of `searching` (:131-154) for a whole batch of result ids I [nq, beam]."""
    tokenizer = tokenizer or SimpleTokenizer()
    metrics, outputs = [], []
    for idx in range(len(I)):
        topk_tbs = [id2doc[str(int(tb_id))] for tb_id in I[idx] if int(tb_id) != -1]   # (-1: fewer rows than beam, tfidffilter:552-555)
        metric_i, output_i = {}, {}
        item = ds_items[idx]
        if eval_only_ans:
            metric_i = {"question": item["question"],
                        "table_recall": int(found_table(item["table_id"], topk_tbs)),
                        "ans_recall": int(para_has_answer(item["answer-text"], topk_tbs, tokenizer)),
                        "type": item.get("type", "single")}
        if with_outputs:
            output_i = {"question_id": item["question_id"], "question": item["question"],
                        "top_{}".format(beam_size): topk_tbs}
            if "answer-text" in item:
                output_i["answer-text"] = item["answer-text"]
            if "table_id" in item:
                output_i["table_id"] = item["table_id"]
        metrics.append(metric_i)
        outputs.append(output_i)
    return metrics, outputs


def get_recall(answers, preds, n=5):
    """scripts/eval_ottqa_retrieval.py:170-174."""
    truth = [any(ans == inst for inst in preds[i][:n]) for i, ans in enumerate(answers)]
    return sum(truth) / len(truth), sum(truth), len(truth)


def table_recall_report(retrieval_outputs, key="top_100"):
    """:176-192 — list of (label, recall, hits, total), labels as the reference prints them."""
    gold = [inst["table_id"] for inst in retrieval_outputs]
    preds = [[item["table_id"] for item in inst[key]] for inst in retrieval_outputs]
    return [(label,) + get_recall(gold, preds, n) for n, label in RECALL_POINTS]


def answer_recall_report(metrics):
    """:194-203."""
    by_type = collections.defaultdict(list)
    for m in metrics:
        by_type[m["type"]].append(m)
    rep = {"ans_recall": float(np.mean([m["ans_recall"] for m in metrics])),
           "table_recall": float(np.mean([m["table_recall"] for m in metrics])), "types": {}}
    for t, items in by_type.items():
        rep["types"][t] = {"num": len(items), "ans_recall": float(np.mean([m["ans_recall"] for m in items]))}
    return rep


def write_jsonl(path, outputs):
    """:163-167."""
    with open(path, "w") as out:
        for rec in outputs:
            out.write(json.dumps(rec) + "\n")


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--raw_data_path", type=str, required=True)
    ap.add_argument("--query_embeddings_path", type=str, required=True)
    ap.add_argument("--corpus_embeddings_path", type=str, required=True)
    ap.add_argument("--id2doc_path", type=str, required=True)
    ap.add_argument("--output_save_path", type=str, default="")
    ap.add_argument("--beam_size", type=int, default=100)
    ap.add_argument("--faiss_save_path", type=str, default="",
                    help="flat IndexFlatIP file: read instead of building when it exists (scripts/eval_ottqa_retrieval.py:106-107)")
    ap.add_argument("--eval_only_ans", action="store_true")
    ap.add_argument("--tfidf_result_file", type=str, default=None,
                    help="{question_id: {'doc_ids': [table ids]}}: search each question only among the blocks of its first "
                         "--topk_doc documents (retrieval/eval_ottqa_retrieval_tfidffilter.py)")
    ap.add_argument("--topk_doc", type=int, default=20)
    ap.add_argument("--gpu", action="store_true",
                    help="every visible GPU, as scripts/eval_ottqa_retrieval.py:110-112 (rows sharded, not replicated); "
                         "without it: GPU 0")
    args = ap.parse_args(argv)
    logging.basicConfig(level=logging.INFO)
    with open(args.raw_data_path, "r", encoding="utf8") as f:
        ds_items = json.load(f)
    q = np.load(args.query_embeddings_path).astype("float32")
    xb = np.load(args.corpus_embeddings_path, mmap_mode="r")
    logger.info("corpus size: %s", (xb.shape,))
    if args.faiss_save_path and os.path.exists(args.faiss_save_path):
        from .index_io import read_index
        index = read_index(args.faiss_save_path)
    else:
        devices = None
        if args.gpu and not args.tfidf_result_file:      # (the filtered script itself uses one GPU, tfidffilter:537-541)
            from . import _lib
            devices = list(range(int(_lib.load().otter_device_count())))
        index = build_index(xb, devices=devices)
    id2doc = load_id2doc(args.id2doc_path)
    if args.tfidf_result_file:
        with open(args.tfidf_result_file, "r", encoding="utf8") as f:
            tfidf_qid2docs = json.load(f)
        D, I = filtered_search(index, q, args.beam_size, tfidf_subsets(ds_items, id2doc, tfidf_qid2docs, args.topk_doc))
    else:
        D, I = index.search(q, args.beam_size)      # one batched call (the reference: nq = 1 per call, 24 processes)
    index.close()
    metrics, outputs = records(ds_items, I, id2doc, args.beam_size, args.eval_only_ans, args.output_save_path != "")
    if args.output_save_path:
        write_jsonl(args.output_save_path, outputs)
        logger.info("Saving outputs to %s", args.output_save_path)
        if "test" not in args.raw_data_path:
            for label, r, t, a in table_recall_report(outputs, "top_{}".format(args.beam_size)):
                logger.info("Table Recall @%s: %s, %s/%s", label, r, t, a)
    if args.eval_only_ans:
        rep = answer_recall_report(metrics)
        logger.info("Ans Recall: %s", rep["ans_recall"])
        logger.info("Table Recall: %s", rep["table_recall"])
        for t, v in rep["types"].items():
            logger.info("%s Questions num: %d", t, v["num"])
            logger.info("Ans Recall: %s", v["ans_recall"])
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
