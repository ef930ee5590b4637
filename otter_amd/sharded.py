"""Row-sharded exact search and data-parallel corpus encoding: one process per GPU,
torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

Replaces the reference's multi-GPU modes:
  * faiss.index_cpu_to_all_gpus (scripts/eval_ottqa_retrieval.py:110-112: replicas of the whole
    index) -> shard g holds rows [g*ceil(N/G), min(N, (g+1)*ceil(N/G))); every rank searches its
    shard for ALL queries (ids already global via id_base), then ONE all-gather of the packed
    (id, score) candidates, [nq, k] per rank -- 8 bytes per candidate: the float32 score and the
    row's 32-bit index INSIDE its shard; the receiver adds the sender's id_base, which every rank
    learnt once at add() -- and a G-way merge on every rank.  When the gathered candidates would be large (config 5: 16,384 queries x top-1000
    = 197 MB per rank, 1.57 GB gathered) the exchange is sliced by queries instead: one all-to-all
    hands rank r every shard's packed candidates for ITS ceil(nq/G) queries (point-to-point
    messages, one xGMI link each, no ring), rank r merges them, and one all-gather of the merged
    packed [nq/G, k] slices follows.
  * nn.DataParallel over the encoder (scripts/encode_corpus.py:112-113: weights re-broadcast and
    outputs gathered every step) -> rank r encodes its own contiguous row range with resident
    weights; no data-path collective (one all-gather of row counts at the end).

The collective itself stays in torch.distributed (plumbing: the process group, its RCCL
communicator and its stream ordering belong to the host program); the C ABI takes and returns
plain device pointers either side of it (otter_index_search into the send buffer,
otter_merge_topk out of the receive buffer).
"""
import numpy as np
import torch
import torch.distributed as dist

NEG = -3.4028234663852886e38   # faiss' "missing" score


def shard_range(n, world, rank):
    """Rows [lo, hi) of shard `rank` (contiguous, ceil(N/G) rows per shard)."""
    per = -(-n // world) if n > 0 else 0
    lo = min(n, rank * per)
    hi = min(n, lo + per)
    return lo, hi


class PackedCandidates:
    """One exchange buffer for `rows` x k candidates: ids first, float32 scores behind them, one
    collective instead of two.  id_bytes = 8: int64 global ids (12 bytes per candidate; a search that
    takes output tensors writes straight into the send buffer); id_bytes = 4: int32 indexes local to
    the sending shard, -1 = missing (8 bytes per candidate).  D and I are views."""

    def __init__(self, rows, k, device, lead=(), id_bytes=8):
        self.rows, self.k, self.id_bytes = int(rows), int(k), int(id_bytes)
        per = self.id_bytes + 4
        self.nbytes = (self.rows * self.k * per + 15) // 16 * 16     # 16-byte multiple: slices of a [G, nbytes] buffer stay aligned
        self.buf = torch.empty(tuple(lead) + (self.nbytes,), dtype=torch.uint8, device=device)

    def views(self, buf=None):
        b = self.buf if buf is None else buf
        cut = self.rows * self.k * self.id_bytes
        lead = b.shape[:-1]
        I = b[..., :cut].view(torch.int64 if self.id_bytes == 8 else torch.int32).view(lead + (self.rows, self.k))
        D = b[..., cut:cut + self.rows * self.k * 4].view(torch.float32).view(lead + (self.rows, self.k))
        return D, I


def to_local_ids(I, base):
    """int64 global ids -> int32 indexes inside the shard starting at `base` (-1 stays -1)."""
    # (valid ids are >= base, the missing -1 falls to -1 - base: two elementwise kernels instead of four)
    return (I - int(base)).clamp_(min=-1).to(torch.int32)


def to_global_ids(I32, bases):
    """[G, rows, k] int32 shard-local indexes -> int64 global ids, bases[g] added to list g (-1 stays -1)."""
    b = bases.to(I32.device).view((-1,) + (1,) * (I32.dim() - 1))
    return torch.where(I32 >= 0, I32 + b, -1)           # (int32 + int64 promotes: no separate widening pass)


class ShardedIndexFlatIP:
    """The IndexFlatIP surface over a corpus sharded by rows across the ranks of a process group.

    index_factory(d, id_base) -> object with add(x)/search(q, k)/ntotal (default: the HIP index on
    this rank's GPU); merge_fn(Dc, Ic) -> (D, I) merges [G, nq, k] candidates (default: the GPU
    merge kernel).  Both are injectable so that the collective protocol is testable on CPU.
    With the default factory queries may be numpy / CPU tensors (results come back as numpy) or
    tensors on this rank's GPU (results stay there).  `exchange_device`: where the collective's
    tensors live -- None = where search() returned them (RCCL: the GPU); "cpu" stages them through
    the host (a gloo group over GPU indexes: bench.py's one-GPU rehearsal)."""

    # candidate bytes over all ranks from which the exchange is sliced by queries ("auto")
    SLICED_FROM_BYTES = 64 << 20

    def __init__(self, d, group=None, device=None, index_factory=None, merge_fn=None, exchange="auto",
                 exchange_device=None, dtype="fp32", compact=True):
        if exchange not in ("auto", "gather", "sliced"):
            raise ValueError("exchange must be 'auto', 'gather' or 'sliced'")
        self.exchange = exchange
        # compact: 8 bytes per candidate on the wire (score + 32-bit shard-local index) instead of 12 (score + int64
        # global id); shards of 2^31 rows or more fall back to 12
        self.compact = bool(compact)
        self.bases = None              # [G] int64: first global id of every rank's shard
        self.exchange_device = None if exchange_device is None else torch.device(exchange_device)
        self.d = int(d)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.device = None
        if index_factory is None:
            from .index import IndexFlatIP, merge_topk
            dev = torch.cuda.current_device() if device is None else torch.device(device).index
            self.device = torch.device("cuda", dev)
            index_factory = lambda dd, base: IndexFlatIP(dd, device=dev, id_base=base, dtype=dtype)  # noqa: E731
            merge_fn = merge_fn or (lambda Dc, Ic: merge_topk(Dc, Ic, device=dev))
        if merge_fn is None:
            raise ValueError("merge_fn is required with a custom index_factory")
        self._factory = index_factory
        self._merge = merge_fn
        self.local = None
        self.ntotal = 0
        self.lo = self.hi = 0
        self.last_exchange = None      # "gather" / "sliced": what the last search() used (tests, bench)
        self.last_exchange_bytes = 0   # bytes this rank sent in the last search()

    def add_global(self, x):
        """Every rank is handed the same [N, d] array (or an object sliceable by rows, e.g. a
        numpy memmap of the reference's embeddings .npy) and keeps only its own rows."""
        if self.local is not None:
            raise RuntimeError("add_global may be called once; use add_shard for incremental loads")
        n = x.shape[0]
        self.lo, self.hi = shard_range(n, self.world, self.rank)
        self.local = self._factory(self.d, self.lo)
        if self.hi > self.lo:
            self.local.add(x[self.lo:self.hi])
        self.ntotal = n
        self.bases = torch.tensor([shard_range(n, self.world, r)[0] for r in range(self.world)], dtype=torch.int64)
        self._max_shard = max(shard_range(n, self.world, r)[1] - shard_range(n, self.world, r)[0] for r in range(self.world))

    def add_shard(self, x_local, id_base, ntotal):
        """This rank's rows are already local (e.g. just written by encode_rows): global ids are
        id_base + row."""
        self.local = self._factory(self.d, int(id_base))
        self.lo, self.hi = int(id_base), int(id_base) + x_local.shape[0]
        if x_local.shape[0]:
            self.local.add(x_local)
        self.ntotal = int(ntotal)
        # every rank learns every shard's first id and size once (the receiver of a compact exchange adds the base)
        mine = torch.tensor([self.lo, self.hi - self.lo], dtype=torch.int64)
        if self.world > 1:
            dev = self.device if (self.device is not None and self.exchange_device is None) else torch.device("cpu")
            allb = torch.empty((self.world * 2,), dtype=torch.int64, device=dev)
            dist.all_gather_into_tensor(allb, mine.to(dev), group=self.group)
            allb = allb.cpu().view(self.world, 2)
        else:
            allb = mine.view(1, 2)
        self.bases = allb[:, 0].contiguous()
        self._max_shard = int(allb[:, 1].max())

    def _local_search(self, q, k):
        """(D, I) torch tensors [nq, k] of this rank's shard, on the device the collective will use."""
        D, I = self.local.search(q, k)
        if not torch.is_tensor(D):
            D, I = torch.from_numpy(np.ascontiguousarray(D)), torch.from_numpy(np.ascontiguousarray(I))
        if self.exchange_device is not None:
            D, I = D.to(self.exchange_device), I.to(self.exchange_device)
        return D, I

    def search(self, q, k):
        """D, I = search(q, k) on every rank (same result everywhere)."""
        if self.local is None:
            raise RuntimeError("search before add")
        as_numpy = False
        if self.device is not None and not (torch.is_tensor(q) and q.is_cuda):
            # host queries with the HIP index: the exchange runs on device tensors (RCCL), numpy comes back
            as_numpy = True
            q = torch.as_tensor(np.ascontiguousarray(q, dtype=np.float32)).to(self.device)
        if self.world == 1:
            D, I = self.local.search(q, k)
        else:
            nq, kk = int(q.shape[0]), int(k)
            mode = self.exchange
            if mode == "auto":
                mode = "sliced" if self.world * nq * kk * 12 >= self.SLICED_FROM_BYTES else "gather"
            self.last_exchange = mode
            compact = self.compact and self._max_shard < (1 << 31)
            if mode == "gather" and not compact and self.device is not None and self.exchange_device is None:
                # the HIP search writes (D, I) straight into the packed send buffer: no pack step
                send = PackedCandidates(nq, kk, self.device)
                Ds, Is = send.views()
                self.local.search(q, kk, out=(Ds, Is))
                D, I = self._gather_packed(send)
            else:
                D, I = self._local_search(q, kk)
                idb = 4 if compact else 8
                D, I = self._exchange_sliced(D, I, idb) if mode == "sliced" else self._exchange_gather(D, I, idb)
            if self.device is not None and D.device != self.device:
                D, I = D.to(self.device), I.to(self.device)
        if as_numpy:
            return D.cpu().numpy(), I.cpu().numpy()
        return D, I

    def _exchange_gather(self, D, I, id_bytes=8):
        """One all-gather of the packed candidates (8 B x nq x k per rank: 1.77 MB at nq = 2214,
        k = 100; 12 B with int64 ids), then the G-way merge on every rank."""
        nq, kk = D.shape
        send = PackedCandidates(nq, kk, D.device, id_bytes=id_bytes)
        Ds, Is = send.views()
        Ds.copy_(D)
        Is.copy_(to_local_ids(I, self.lo) if id_bytes == 4 else I)
        return self._gather_packed(send)

    def _gather_packed(self, send):
        G = self.world
        recv = torch.empty((G * send.nbytes,), dtype=torch.uint8, device=send.buf.device)
        dist.all_gather_into_tensor(recv, send.buf, group=self.group)
        self.last_exchange_bytes = send.nbytes
        Dg, Ig = send.views(recv.view(G, send.nbytes))
        if send.id_bytes == 4:
            Ig = to_global_ids(Ig, self._bases_on(Ig.device))
        return self._merged(Dg, Ig)

    def _bases_on(self, device):
        """The shards' first ids as a tensor on `device`, copied there once (not per search)."""
        cache = getattr(self, "_bases_cache", None)
        if cache is None or cache[0] is not self.bases or cache[1].device != device:
            cache = (self.bases, self.bases.to(device))
            self._bases_cache = cache
        return cache[1]

    def _merged(self, Dg, Ig):
        dev = Dg.device
        if self.device is not None and dev != self.device:
            Dg, Ig = Dg.to(self.device), Ig.to(self.device)
        D, I = self._merge(Dg.contiguous(), Ig.contiguous())
        if not torch.is_tensor(D):
            D, I = torch.from_numpy(D), torch.from_numpy(I)
        return D, I

    def _exchange_sliced(self, D, I, id_bytes=8):
        """SURVEY.md §8e option B.  Rank r owns queries [r*per, (r+1)*per), per = ceil(nq/G); rows past
        nq are padding (score -FLT_MAX, id -1: the "missing" convention the merge already handles).
        The all-to-all carries this shard's candidates (shard-local 32-bit indexes when id_bytes = 4: every
        received list g gets rank g's base added); the all-gather of the merged slices carries global ids."""
        G = self.world
        nq, kk = D.shape
        per = -(-nq // G)
        pad = per * G
        if id_bytes == 4:
            I = to_local_ids(I, self.lo)
        # chunk r of the send buffer = packed candidates of rank r's queries
        send = PackedCandidates(per, kk, D.device, lead=(G,), id_bytes=id_bytes)
        Ds, Is = send.views()
        for r in range(G):
            lo, hi = r * per, min(nq, (r + 1) * per)
            if hi > lo:
                Ds[r, :hi - lo].copy_(D[lo:hi])
                Is[r, :hi - lo].copy_(I[lo:hi])
            if hi - lo < per:
                Ds[r, max(0, hi - lo):].fill_(NEG)
                Is[r, max(0, hi - lo):].fill_(-1)
        recv = torch.empty_like(send.buf)
        dist.all_to_all_single(recv.view(-1), send.buf.view(-1), group=self.group)
        Dr, Ir = send.views(recv)                              # [G, per, kk]: every shard's list of MY queries
        if id_bytes == 4:
            Ir = to_global_ids(Ir, self._bases_on(Ir.device))
        Dm, Im = self._merged(Dr, Ir)
        mine = PackedCandidates(per, kk, D.device)
        Dv, Iv = mine.views()
        Dv.copy_(Dm.to(D.device))
        Iv.copy_(Im.to(I.device))
        allb = torch.empty((G * mine.nbytes,), dtype=torch.uint8, device=D.device)
        dist.all_gather_into_tensor(allb, mine.buf, group=self.group)
        self.last_exchange_bytes = send.buf.numel() + mine.nbytes
        Dg, Ig = mine.views(allb.view(G, mine.nbytes))         # [G, per, kk] in query order
        return Dg.reshape(pad, kk)[:nq].contiguous(), Ig.reshape(pad, kk)[:nq].contiguous()


def encode_rows(model, batches, n_rows, group=None, to_cpu=False):
    """Data-parallel corpus encoding.  `batches(lo, hi)` yields the batch dicts covering rows
    [lo, hi) of the corpus in order; this rank encodes only its shard_range.  Returns
    (embeds [hi-lo, D], lo, counts) where counts[r] is the number of rows rank r produced
    (the single collective).  A failing batch raises: rows are never dropped (contrast
    scripts/encode_corpus.py:138-143, which would shift every later row id)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lo, hi = shard_range(n_rows, world, rank)
    outs = []
    for batch in batches(lo, hi):
        e = model(batch)["embed"]
        outs.append(e.cpu() if to_cpu else e)
    if outs:
        embeds = torch.cat(outs, dim=0)
    else:
        embeds = torch.empty((0, 0))
    if embeds.shape[0] != hi - lo:
        raise RuntimeError("rank %d encoded %d rows, expected %d" % (rank, embeds.shape[0], hi - lo))
    counts = [hi - lo]
    if world > 1:
        dev = embeds.device if embeds.is_cuda else torch.device("cpu")
        mine = torch.tensor([hi - lo], dtype=torch.int64, device=dev)
        allc = torch.empty((world,), dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(allc, mine, group=group)
        counts = [int(c) for c in allc.cpu()]
        if sum(counts) != n_rows:
            raise RuntimeError("shards cover %d rows, corpus has %d" % (sum(counts), n_rows))
    return embeds, lo, counts
