"""Host loop of corpus / question encoding: the counterpart of predict() and the np.save in
reference scripts/encode_corpus.py:116-151.

    embeds = predict(model, eval_dataloader)          # [N, D] float32, dataset order
    save_embeddings(path, embeds)                     # np.save, row i <-> id2doc[str(i)]

Unlike the reference (its `try/except ... continue`, :138-143) a failing batch raises: dropping a
batch would silently shift every later row against id2doc.json
(retrieval/data/encode_datasets.py:145-152).
"""
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
            src = t.to(torch.int64).contiguous() if t.dtype != torch.int64 else t.contiguous()
            if not self.use_cuda:
                out[name] = src
                continue
            n = src.numel()
            pin = self._flat(self.pin, name, n, torch.int64, True)[:n].view(src.shape)
            # (a plain memcpy: torch's copy_ fans 512 KB out over its thread pool, whose wake-up after an idle
            # phase was measured at up to 32 ms -- with nothing queued on the GPU behind it)
            np.copyto(pin.numpy(), src.numpy())
            devt = self._flat(self.dev, name, n, torch.int64, False, device)[:n].view(src.shape)
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
