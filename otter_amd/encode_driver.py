"""Host loop of corpus / question encoding: the counterpart of predict() and the np.save in
reference scripts/encode_corpus.py:116-151.

    embeds = predict(model, eval_dataloader)          # [N, D] float32, dataset order
    save_embeddings(path, embeds)                     # np.save, row i <-> id2doc[str(i)]

Unlike the reference (its `try/except ... continue`, :138-143) a failing batch raises: dropping a
batch would silently shift every later row against id2doc.json
(retrieval/data/encode_datasets.py:145-152).
"""
import json
import os

import numpy as np
import torch

from .encoder import move_to_cuda  # noqa: F401  (re-exported: the reference's predict() calls it per batch)


class _PipeSlot:
    """One of the two in-flight batches of predict(): pinned host staging for the token tensors, their
    device copies, a pinned host buffer for the embeddings and the events that order the three streams."""

    def __init__(self, use_cuda):
        self.use_cuda = use_cuda
        self.pin = {}            # name -> flat pinned int64 buffer (grow-only)
        self.dev = {}            # name -> flat device int64 buffer (grow-only)
        self.out_pin = None      # flat pinned float32 buffer
        self.out_dev = None      # the forward's result, kept alive until its download ran
        self.rows = 0
        self.dim = 0
        self.busy = False
        self.ran = False         # a forward has read this slot's device inputs (ev_fwd is recorded)
        if use_cuda:
            self.ev_h2d = torch.cuda.Event()
            self.ev_fwd = torch.cuda.Event()
            self.ev_d2h = torch.cuda.Event()

    def _flat(self, store, name, numel, dtype, pinned, device=None):
        t = store.get(name)
        if t is None or t.numel() < numel:
            if pinned:
                t = torch.empty((numel,), dtype=dtype, pin_memory=self.use_cuda)
            else:
                t = torch.empty((numel,), dtype=dtype, device=device)
            store[name] = t
        return t

    def upload(self, batch, device, copy_stream):
        """Host batch dict -> device batch dict; the copies are enqueued on copy_stream."""
        if self.use_cuda:
            self.ev_h2d.synchronize()     # the pinned staging of this slot's previous batch has been read
            if self.ran:                  # ... and its device copies by the forward that used them (device-side wait)
                copy_stream.wait_event(self.ev_fwd)
        out = {}
        for name, t in batch.items():
            if not torch.is_tensor(t):
                out[name] = t
                continue
            if t.is_cuda:                 # already resident: nothing to stage
                out[name] = t
                continue
            # integer and bool tensors (ids, masks) travel as int64, which is what the encoder takes; anything else
            # keeps its own dtype, as move_to_cuda does (retrieval/utils/utils.py:31-48) -- never a silent truncation
            if t.dtype == torch.int64 or t.is_floating_point() or t.is_complex():
                src = t.contiguous()
            else:
                src = t.to(torch.int64).contiguous()
            if not self.use_cuda:
                out[name] = src
                continue
            n = src.numel()
            key = (name, src.dtype)
            pin = self._flat(self.pin, key, n, src.dtype, True)[:n].view(src.shape)
            # (a plain memcpy: torch's copy_ fans 512 KB out over its thread pool, whose wake-up after an idle
            # phase was measured at up to 32 ms -- with nothing queued on the GPU behind it)
            np.copyto(pin.numpy(), src.numpy())
            devt = self._flat(self.dev, key, n, src.dtype, False, device)[:n].view(src.shape)
            with torch.cuda.stream(copy_stream):
                devt.copy_(pin, non_blocking=True)
            out[name] = devt
        if self.use_cuda:
            self.ev_h2d.record(copy_stream)
        return out

    def download(self, embed, down_stream, to_cpu):
        """Enqueue the device -> pinned host copy of one batch's embeddings behind its forward."""
        self.rows, self.dim = int(embed.shape[0]), int(embed.shape[1])
        self.out_dev = embed
        self.busy = True
        if self.use_cuda and embed.is_cuda:
            self.ev_fwd.record(torch.cuda.current_stream(embed.device))
            self.ran = True
        if not (to_cpu and self.use_cuda and embed.is_cuda):
            return
        n = embed.numel()
        if self.out_pin is None or self.out_pin.numel() < n:
            self.out_pin = torch.empty((n,), dtype=torch.float32, pin_memory=True)
        with torch.cuda.stream(down_stream):
            down_stream.wait_event(self.ev_fwd)
            self.out_pin[:n].view(self.rows, self.dim).copy_(embed, non_blocking=True)
            self.ev_d2h.record(down_stream)

    def collect(self, to_cpu, sink=None):
        """The finished batch's embeddings (host copy when to_cpu); frees the slot.  `sink(rows, dim)` may hand out
        the rows of a preallocated result to copy into (no per-batch allocation, no concatenation at the end)."""
        emb = self.out_dev
        if to_cpu and self.use_cuda and emb.is_cuda:
            self.ev_d2h.synchronize()
            src = self.out_pin[:self.rows * self.dim].view(self.rows, self.dim)
            dst = sink(self.rows, self.dim) if sink is not None else None
            if dst is None:
                dst = torch.empty((self.rows, self.dim), dtype=torch.float32)
            np.copyto(dst.numpy(), src.numpy())      # (memcpy, see upload)
            emb = dst
        elif to_cpu:
            emb = emb.cpu()
        self.out_dev = None
        self.busy = False
        return emb


_PIPE_CACHE = {}


def predict(model, eval_dataloader, to_cpu=True, device=None):
    """scripts/encode_corpus.py:126-151 without its per-batch stalls.  The reference moves a batch to the GPU,
    runs the model and calls `.cpu()` on the result before it touches the next batch.  Here two batches are in
    flight (the side-stream idea of the reference's own training prefetcher, retrieval/data/pre_dataset.py:28-61):
    while batch i runs on the compute stream, batch i + 1's token tensors go from pinned host staging to the
    device on a copy stream and batch i - 1's embeddings go to pinned host memory on a download stream.
    Row order = dataloader order; a failing batch raises (rows are never dropped)."""
    model.eval()
    use_cuda = torch.cuda.is_available()
    if use_cuda:
        if device is None:
            idx = getattr(model, "device_index", None)
            device = torch.device("cuda", torch.cuda.current_device() if idx is None else idx)
        device = torch.device(device)
        # streams and pinned staging are kept per device between calls (pinned allocations cost milliseconds)
        if device not in _PIPE_CACHE:
            _PIPE_CACHE[device] = (torch.cuda.Stream(device=device), torch.cuda.Stream(device=device),
                                   [_PipeSlot(True), _PipeSlot(True)])
        copy_stream, down_stream, slots = _PIPE_CACHE[device]
        cur = torch.cuda.current_stream(device)
        for s in slots:                  # a call that raised half way must not leave its batches to the next one
            if s.busy:
                if s.ran:
                    s.ev_fwd.synchronize()
                s.out_dev, s.busy = None, False
    else:
        copy_stream = down_stream = cur = None
        slots = [_PipeSlot(False), _PipeSlot(False)]
    embed_array = []
    # a loader that knows its length lets the host result be allocated once (rows of the first batch x batches is an
    # upper bound: only the last batch of a loader is short); every batch is then copied into place while the GPU
    # works, instead of one concatenation of the whole result after the last batch
    nbatches = len(eval_dataloader) if hasattr(eval_dataloader, "__len__") else None
    whole = {"buf": None, "rows": 0}

    def sink(rows, dim):
        if nbatches is None or not (to_cpu and use_cuda):
            return None
        if whole["buf"] is None:
            whole["buf"] = torch.empty((rows * nbatches, dim), dtype=torch.float32)
        if whole["buf"].shape[1] != dim or whole["rows"] + rows > whole["buf"].shape[0]:
            return None                  # (a loader whose batches grow: fall back to the list)
        lo = whole["rows"]
        whole["rows"] = lo + rows
        return whole["buf"][lo:lo + rows]

    it = iter(eval_dataloader)

    def stage(i):
        try:
            batch = next(it)
        except StopIteration:
            return None
        s = slots[i % 2]
        if s.busy:                       # its previous batch (i - 2) has not been collected yet
            embed_array.append(s.collect(to_cpu, sink))
        return s.upload(batch, device, copy_stream)

    i = 0
    nxt = stage(0)
    pending = []                         # slots whose embeddings are on their way, oldest first
    while nxt is not None:
        s = slots[i % 2]
        if use_cuda:
            cur.wait_event(s.ev_h2d)
        embed = model(nxt)["embed"]
        s.download(embed, down_stream, to_cpu)
        pending.append(s)
        # stage batch i + 1 while batch i computes; its slot holds batch i - 1, collected first
        if len(pending) == 2:
            embed_array.append(pending.pop(0).collect(to_cpu, sink))
        i += 1
        nxt = stage(i)
    for s in pending:
        embed_array.append(s.collect(to_cpu, sink))
    if not embed_array:
        raise ValueError("predict: the dataloader produced no batches")
    buf = whole["buf"]
    if buf is not None and all(e.untyped_storage().data_ptr() == buf.untyped_storage().data_ptr() for e in embed_array):
        return buf[:whole["rows"]]       # every batch went into the preallocated result, in order
    return torch.cat(embed_array, dim=0)


def save_embeddings(path, embeds):
    """np.save(args.embed_save_path, embeds.cpu().numpy()) — scripts/encode_corpus.py:119-123."""
    d = os.path.dirname(path)
    if d and not os.path.exists(d):
        os.makedirs(d)
    np.save(path, embeds.detach().cpu().numpy().astype(np.float32, copy=False))


def collate_tokens(values, pad_idx):
    """retrieval/data/data_utils.py:5-31 (1-d case used by em_collate_bert): right-pad to the batch max."""
    values = [v.view(-1) for v in values]
    size = max(v.size(0) for v in values)
    res = values[0].new_full((len(values), size), pad_idx)
    for i, v in enumerate(values):
        res[i, : v.numel()] = v
    return res


def em_collate_bert(samples, pad_id=0):
    """retrieval/data/encode_datasets.py:393-417: list of dataset items ({'tb': tokenizer features})
    -> batch dict {input_ids, input_mask[, part2_mask, part3_mask, input_type_ids]}."""
    if len(samples) == 0:
        return {}
    batch = {
        "input_ids": collate_tokens([s["tb"]["input_ids"].view(-1) for s in samples], pad_id),
        "input_mask": collate_tokens([s["tb"]["attention_mask"].view(-1) for s in samples], 0),
    }
    for src, dst in (("part2_mask", "part2_mask"), ("part3_mask", "part3_mask"), ("token_type_ids", "input_type_ids")):
        if src in samples[0]["tb"]:
            batch[dst] = collate_tokens([s["tb"][src].view(-1) for s in samples], 0)
    return batch


# ------------------------------------------------------------------ the script: scripts/encode_corpus.py:25-123
def load_records(path, table, allow_pickle=False):
    """The reference reads questions from JSON and the corpus from a pickle of records whose `table` is a pandas
    DataFrame (retrieval/data/encode_datasets.py:128-136).  Here: .json (a list) or .jsonl (one record per line; a
    record's `table` is then [header, [[row]]], the form id2doc.json holds); a .pkl corpus is read only on request
    (allow_pickle: unpickling executes what the file says, so it must be the caller's own file)."""
    ext = os.path.splitext(path)[1].lower()
    if ext in (".pkl", ".pickle"):
        if not allow_pickle:
            raise ValueError("%s is a pickle; pass --allow_pickle for a file you wrote yourself, or convert it to .jsonl" % path)
        import pickle
        with open(path, "rb") as f:
            return pickle.load(f)
    with open(path, "r", encoding="utf-8") as f:
        if ext == ".jsonl":
            return [json.loads(line) for line in f if line.strip()]
        return json.load(f)


def build_model(bert_config, args, dtype=None):
    """The class choice of scripts/encode_corpus.py:48-69."""
    from . import encoder as E
    roberta = "roberta" in args.model_name
    if args.three_cat:
        if args.part_pooling[:3] == "att":
            cls = E.RobertaSingleEncoderThreeCatAtt if roberta else E.SingleEncoderThreeCatAtt
        else:
            cls = E.RobertaSingleEncoderThreeCatPool if roberta else E.SingleEncoderThreeCatPool
    else:
        cls = E.RobertaSingleEncoder if roberta else E.SingleEncoder
    return cls(bert_config, args, dtype=dtype)


def encode_args(argv=None):
    """The arguments of retrieval/config.py:encode_args that the encode path reads (same names and defaults;
    --metadata / --three_cat / --no_proj / --normalize_table are always-true switches there, kept so)."""
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--predict_file", type=str, required=True)
    ap.add_argument("--model_name", type=str, default="roberta-base", help="HF name or local directory (config + tokenizer)")
    ap.add_argument("--init_checkpoint", type=str, required=True)
    ap.add_argument("--embed_save_path", type=str, required=True)
    ap.add_argument("--encode_table", action="store_true")
    ap.add_argument("--predict_batch_size", type=int, default=512)
    ap.add_argument("--num_workers", type=int, default=0)
    ap.add_argument("--max_c_len", type=int, default=512)
    ap.add_argument("--max_q_len", type=int, default=70)
    ap.add_argument("--psg_mode", type=str, default="ori")
    ap.add_argument("--no_passages", action="store_true")
    ap.add_argument("--fp16", action="store_true", help="bf16 weights and activations (the reference: apex amp O1)")
    ap.add_argument("--shared_encoder", action="store_true")
    ap.add_argument("--part_pooling", type=str, default="first")
    ap.add_argument("--metadata", action="store_true", default=True)
    ap.add_argument("--three_cat", action="store_true", default=True)
    ap.add_argument("--no_proj", action="store_true", default=True)
    ap.add_argument("--normalize_table", action="store_true", default=True)
    ap.add_argument("--one_query", action="store_true", default=True)
    ap.add_argument("--one_vector", dest="three_cat", action="store_false", help="768-d encoders (RobertaSingleEncoder / SingleEncoder)")
    ap.add_argument("--no_cuda", action="store_true", help="refused: there is no CPU path")
    ap.add_argument("--local_rank", type=int, default=-1)
    ap.add_argument("--allow_pickle", action="store_true")
    return ap.parse_args(argv)


def main(argv=None):
    """python -m otter_amd.encode_driver ... = scripts/encode_corpus.py: tokenizer and config from --model_name,
    EmDatasetMetaThreeCat (EmDatasetMeta with --one_vector) over --predict_file, the encoder class the flags select, load_saved(--init_checkpoint),
    every visible GPU from one process (encoder.DataParallel) unless launched one process per GPU (RANK / WORLD_SIZE
    set: sharded.encode_rows, one shard file per rank), predict(), np.save(--embed_save_path) and, for --encode_table,
    <embed_save_path>/id2doc.json (retrieval/data/encode_datasets.py:141-154)."""
    import logging
    from functools import partial

    from torch.utils.data import DataLoader
    from transformers import AutoConfig, AutoTokenizer

    from . import encoder as E
    from . import features as F
    logging.basicConfig(level=logging.INFO)
    log = logging.getLogger("otter_amd.encode_driver")
    args = encode_args(argv)
    if args.no_cuda or not torch.cuda.is_available():
        raise RuntimeError("otter_amd encodes on the GPU only (no CPU fallback)")
    bert_config = AutoConfig.from_pretrained(args.model_name)
    tokenizer = AutoTokenizer.from_pretrained(args.model_name)
    collate_fc = partial(em_collate_bert, pad_id=tokenizer.pad_token_id)
    model = build_model(bert_config, args)
    log.info("Model Using %s...", type(model).__name__)
    data = load_records(args.predict_file, args.encode_table, args.allow_pickle)
    log.info("Total sample count %d", len(data))
    Dataset = F.EmDatasetMetaThreeCat if args.three_cat else F.EmDatasetMeta      # scripts/encode_corpus.py:80-86
    model = E.load_saved(model, args.init_checkpoint, exact=False)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        import torch.distributed as dist

        from . import sharded
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        if not dist.is_initialized():
            dist.init_process_group(backend="nccl")
        model.to("cuda:%d" % local)

        def batches(lo, hi):
            ds = Dataset(tokenizer, data[lo:hi], args.encode_table, args)
            return DataLoader(ds, batch_size=args.predict_batch_size, collate_fn=collate_fc, num_workers=args.num_workers)
        embeds, lo, counts = sharded.encode_rows(model, batches, len(data), to_cpu=True)
        F.write_embedding_shards(args.embed_save_path, embeds, lo, world, dist.get_rank())
        log.info("rank %d wrote rows [%d, %d) of %d", dist.get_rank(), lo, lo + embeds.shape[0], len(data))
        if dist.get_rank() == 0 and args.encode_table:
            F.write_id2doc(os.path.join(args.embed_save_path, "id2doc.json"), data)
        return 0
    n_gpu = torch.cuda.device_count()
    model.to("cuda:0")
    if n_gpu > 1:
        model = E.DataParallel(model)
    ds = Dataset(tokenizer, data, args.encode_table, args)
    loader = DataLoader(ds, batch_size=args.predict_batch_size, collate_fn=collate_fc, pin_memory=True,
                        num_workers=args.num_workers)
    if args.encode_table:
        id2doc_path = os.path.join(args.embed_save_path, "id2doc.json")
        if not os.path.exists(id2doc_path):
            F.write_id2doc(id2doc_path, data)
            log.info("Dumping id2doc to %s", id2doc_path)
    embeds = predict(model, loader)
    log.info("%s", tuple(embeds.shape))
    log.info("saving to :%s", args.embed_save_path)
    save_embeddings(args.embed_save_path if args.embed_save_path.endswith(".npy") else args.embed_save_path + ".npy", embeds)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
