"""Feature builder and on-disk formats either side of the encoder: host-side
restatement of the reference's text -> batch path, producing exactly the int64 tensors the encoder's
forward consumes.

    table / passage string templates   retrieval/data/tr_dataset.py:29-46
    get_passages                       utils/common.py:151-161
    three-part mask rule               retrieval/data/tr_dataset.py:117-153 (question branch :125-133,
                                       block branch :139-149)
    dataset item                       retrieval/data/encode_datasets.py:228-256 (<= 8 passages, '?' stripped);
                                       :191-226 + tr_dataset.py:58-81 for the 768-d encoders
    id2doc.json                        retrieval/data/encode_datasets.py:141-154
    corpus record                      preprocessing/corpus_preprocess.py:203-221
    collate                            otter_amd.encode_driver.em_collate_bert (encode_datasets.py:393-417)

The tokenizer is whatever HF tokenizer the caller loads (AutoTokenizer.from_pretrained(args.model_name),
scripts/encode_corpus.py:46); nothing here depends on a particular vocabulary.  CPU string work only:
no kernels, no device memory.
"""
import json
import os

import numpy as np
import torch


# ------------------------------------------------------------------ string templates
def _header_values(table):
    """A one-row pandas DataFrame (the reference's record) or a [header, [[row]]] pair (id2doc.json)."""
    if hasattr(table, "columns"):
        return table.columns.tolist(), table.values.tolist()
    return list(table[0]), [list(r) for r in table[1]]


def table_is_empty(table):
    if table is None:
        return True
    if hasattr(table, "empty"):
        return bool(table.empty)
    return len(table) == 0 or len(table[0]) == 0


def convert_tb_to_string_metadata(table, passages, meta_data):
    """retrieval/data/tr_dataset.py:29-35."""
    header, value = _header_values(table)
    table_str = " [TITLE] " + meta_data["title"] + " [SECTITLE] " + meta_data["section_title"] + \
                " [HEADER] " + " [SEP] ".join(header) + " [DATA] " + " [SEP] ".join(value[0])
    passage_str = " [PASSAGE] " + " [SEP] ".join(passages)
    return table_str, passage_str


def convert_tb_to_string_metadata_norm(table, passages, meta_data):
    """retrieval/data/tr_dataset.py:37-46 ("H1 is C1." form; the default: normalize_table=True)."""
    header, value = _header_values(table)
    table_str = " [TAB] " + " [TITLE] " + meta_data["title"] + " [SECTITLE] " + meta_data["section_title"] + \
                " [DATA] " + " ".join(["{} is {}.".format(h, c) for h, c in zip(header, value[0])])
    passage_str = " [PASSAGE] " + " [SEP] ".join(passages)
    return table_str, passage_str


def get_passages(js, psg_mode, neg=False):
    """utils/common.py:151-161."""
    prefix = "neg_" if neg else ""
    if psg_mode == "ori":
        return js[prefix + "passages"]
    if psg_mode == "s_sent":
        return js[prefix + "s_sent"] if len(js[prefix + "s_sent"]) > 0 else js[prefix + "passages"]
    if psg_mode == "s_psg":
        return js[prefix + "s_psg"] if len(js[prefix + "s_psg"]) > 0 else js[prefix + "passages"]
    return []


def _encode_plus(tokenizer, text, text_pair=None, **kw):
    """tokenizer.encode_plus of transformers 4.x (what the reference calls); __call__ where 5.x dropped it."""
    fn = getattr(tokenizer, "encode_plus", None)
    if fn is None:
        return tokenizer(text, text_pair=text_pair, **kw)
    return fn(text, text_pair=text_pair, **kw)


# ------------------------------------------------------------------ mask rule
def threecat_masks(special_tokens_mask, question):
    """part2 / part3 masks of one tokenised item (retrieval/data/tr_dataset.py:125-149).

    zp = positions where special_tokens_mask == 1 (CLS, every SEP and every PAD).
      part2 = 1 on [1, zp[1])
      block:    part3 = 1 on (zp[2], zp[3])
      question: part3 = part2
    """
    stm = np.asarray(special_tokens_mask).reshape(-1)
    n = stm.shape[0]
    zp = np.nonzero(stm == 1)[0]
    part2 = np.zeros(n, np.int64)
    part2[1:int(zp[1])] = 1
    if question:
        return part2, part2.copy()
    part3 = np.zeros(n, np.int64)
    part3[int(zp[2]) + 1:int(zp[3])] = 1
    return part2, part3


def convert_tb_to_features_bert_metadata_threecat_one_query(passages, table, meta_data, tokenizer, args, encode=False):
    """retrieval/data/tr_dataset.py:117-153: dict of [1, L] int64 tensors
    {input_ids, attention_mask[, token_type_ids], length, part2_mask, part3_mask}."""
    if table_is_empty(table):
        if isinstance(passages, str):
            passage_str = passages
        elif isinstance(passages, list):
            passage_str = " ".join(passages)
        else:
            raise TypeError("type passages: {}, {}, {}, {}".format(type(passages), passages, table, meta_data))
        inputs = _encode_plus(tokenizer, passage_str, max_length=args.max_q_len, add_special_tokens=True,
                                       padding="max_length", return_tensors="pt", truncation=True,
                                       return_special_tokens_mask=True, return_length=True)
        question = True
    else:
        if getattr(args, "normalize_table", True):
            table_str, passage_str = convert_tb_to_string_metadata_norm(table, passages, meta_data)
        else:
            table_str, passage_str = convert_tb_to_string_metadata(table, passages, meta_data)
        inputs = _encode_plus(tokenizer, table_str, text_pair=passage_str, max_length=args.max_c_len,
                                       add_special_tokens=True, padding="max_length", return_tensors="pt",
                                       truncation=True, return_special_tokens_mask=True, return_length=True)
        question = False
    out = {k: v for k, v in inputs.items()}
    p2, p3 = threecat_masks(out["special_tokens_mask"].numpy(), question)
    out["part2_mask"] = torch.from_numpy(p2)[None]
    out["part3_mask"] = torch.from_numpy(p3)[None]
    return out


def convert_tb_to_features_bert_metadata(passages, table, meta_data, tokenizer, args, encode=False):
    """retrieval/data/tr_dataset.py:58-81 (the 768-d encoders' features: no part masks):
    {input_ids, attention_mask[, token_type_ids], length}, [1, L] int64 each."""
    if table_is_empty(table):
        if isinstance(passages, str):
            passage_str = passages
        elif isinstance(passages, list):
            passage_str = " ".join(passages)
        else:
            raise TypeError("type passages: {}, {}, {}, {}".format(type(passages), passages, table, meta_data))
        inputs = _encode_plus(tokenizer, passage_str, max_length=args.max_q_len, add_special_tokens=True,
                              padding="max_length", return_tensors="pt", truncation=True, return_length=True)
    else:
        if getattr(args, "normalize_table", True):
            table_str, passage_str = convert_tb_to_string_metadata_norm(table, passages, meta_data)
        else:
            table_str, passage_str = convert_tb_to_string_metadata(table, passages, meta_data)
        inputs = _encode_plus(tokenizer, table_str, text_pair=passage_str, max_length=args.max_c_len,
                              add_special_tokens=True, padding="max_length", return_tensors="pt", truncation=True,
                              return_length=True)
    return {k: v for k, v in inputs.items()}


# ------------------------------------------------------------------ dataset
class EmDatasetMetaThreeCat:
    """retrieval/data/encode_datasets.py:228-256 (+ processing_data :128-155) without pickle: the corpus is
    a list of records {table_id, row_id, table, passages, passages_id, meta_data, ...}
    (preprocessing/corpus_preprocess.py:203-221) handed in by the caller, or read from JSON lines."""

    def __init__(self, tokenizer, data, table, args):
        self.tokenizer = tokenizer
        self.data = data
        self.table = table          # True: table-text blocks, False: questions
        self.args = args

    def __len__(self):
        return len(self.data)

    def __getitem__(self, index):
        js = self.data[index]
        if self.table:
            if getattr(self.args, "no_passages", False):
                psg = [" ", " "]
            else:
                psg = get_passages(js, getattr(self.args, "psg_mode", "ori"), neg=False)[:8]
            tb = convert_tb_to_features_bert_metadata_threecat_one_query(psg, js["table"], js["meta_data"],
                                                                         tokenizer=self.tokenizer, args=self.args)
        else:
            question = js["question"]
            if question.endswith("?"):
                question = question[:-1]
            tb = convert_tb_to_features_bert_metadata_threecat_one_query(question, None, meta_data=None,
                                                                         tokenizer=self.tokenizer, args=self.args)
        return {"tb": tb}


class EmDatasetMeta(EmDatasetMetaThreeCat):
    """retrieval/data/encode_datasets.py:191-226: the dataset of the 768-d encoders (RobertaSingleEncoder /
    SingleEncoder, scripts/encode_corpus.py:84-86): same strings, <= 8 passages, no part masks."""

    def __getitem__(self, index):
        js = self.data[index]
        if self.table:
            psg = get_passages(js, getattr(self.args, "psg_mode", "ori"), neg=False)[:8]
            tb = convert_tb_to_features_bert_metadata(psg, js["table"], js["meta_data"], tokenizer=self.tokenizer,
                                                      args=self.args)
        else:
            question = js["question"]
            if question.endswith("?"):
                question = question[:-1]
            tb = convert_tb_to_features_bert_metadata(question, None, meta_data=None, tokenizer=self.tokenizer,
                                                      args=self.args)
        return {"tb": tb}


# ------------------------------------------------------------------ on-disk formats
def id2doc_from_corpus(data):
    """retrieval/data/encode_datasets.py:141-152: row index -> block record (keys become strings in JSON)."""
    id2doc = {}
    for idx, doc in enumerate(data):
        header, values = _header_values(doc["table"])
        id2doc[idx] = {"table_id": doc["table_id"], "row_id": doc["row_id"], "table": [header, values],
                       "meta_data": doc.get("meta_data", None), "passages_id": doc["passages_id"],
                       "passages": doc["passages"]}
    return id2doc


def write_id2doc(path, data):
    """json.dump(id2doc, g, indent=1) — encode_datasets.py:153-154."""
    d = os.path.dirname(path)
    if d and not os.path.exists(d):
        os.makedirs(d)
    with open(path, "w") as g:
        json.dump(id2doc_from_corpus(data), g, indent=1)


def open_embeddings(path):
    """The reference's embedding matrix (np.save at scripts/encode_corpus.py:123): float [N, D] .npy,
    row i <-> id2doc[str(i)].  Memory-mapped, so IndexFlatIP.add can stream it slab by slab."""
    x = np.load(path, mmap_mode="r", allow_pickle=False)
    if x.ndim != 2:
        raise ValueError("%s: expected a [N, D] matrix, got shape %r" % (path, x.shape))
    return x


def write_embedding_shards(prefix, embeds, lo, world, rank):
    """Row-sharded store written by the data-parallel encode (otter_amd.sharded.encode_rows): one .npy per rank
    plus a small JSON manifest; open_embedding_shards() maps them back in row order."""
    d = os.path.dirname(prefix)
    if d and not os.path.exists(d):
        os.makedirs(d)
    arr = embeds.detach().cpu().numpy() if torch.is_tensor(embeds) else np.asarray(embeds)
    np.save("%s.shard%05d.npy" % (prefix, rank), arr.astype(np.float32, copy=False))
    with open("%s.shard%05d.json" % (prefix, rank), "w") as f:
        json.dump({"rank": rank, "world": world, "row_begin": int(lo), "rows": int(arr.shape[0]),
                   "dim": int(arr.shape[1]) if arr.ndim == 2 else 0}, f)


def open_embedding_shards(prefix):
    """[(row_begin, memmap [rows, D])] in row order; checks that the shards tile [0, N) exactly."""
    import glob
    metas = []
    for mp in sorted(glob.glob(prefix + ".shard*.json")):
        with open(mp) as f:
            metas.append((json.load(f), mp[:-5] + ".npy"))
    if not metas:
        raise FileNotFoundError("no shards at %s.shard*.json" % prefix)
    metas.sort(key=lambda m: m[0]["row_begin"])
    out, expect = [], 0
    for meta, npy in metas:
        if meta["row_begin"] != expect:
            raise ValueError("shard %s starts at row %d, expected %d" % (npy, meta["row_begin"], expect))
        x = np.load(npy, mmap_mode="r", allow_pickle=False)
        if x.shape[0] != meta["rows"]:
            raise ValueError("shard %s holds %d rows, manifest says %d" % (npy, x.shape[0], meta["rows"]))
        out.append((meta["row_begin"], x))
        expect += meta["rows"]
    return out
