"""Row-sharded exact search and data-parallel corpus encoding: one process per GPU,
torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

Replaces the reference's multi-GPU modes:
  * faiss.index_cpu_to_all_gpus (scripts/eval_ottqa_retrieval.py:110-112: replicas of the whole
    index) -> shard g holds rows [g*ceil(N/G), min(N, (g+1)*ceil(N/G))); every rank searches its
    shard for ALL queries (ids already global via id_base), then ONE all-gather of the
    (score, id) candidates [nq, k] per rank and a G-way merge on every rank.  When the gathered
    candidates would be large (config 5: 16,384 queries x top-1000 = 197 MB per rank, 1.57 GB
    gathered) the exchange is sliced by queries instead: an all-to-all hands rank r every shard's
    candidates for ITS ceil(nq/G) queries (point-to-point messages, one xGMI link each, no ring),
    rank r merges them, and one all-gather of the merged [nq/G, k] slices follows.
  * nn.DataParallel over the encoder (scripts/encode_corpus.py:112-113: weights re-broadcast and
    outputs gathered every step) -> rank r encodes its own contiguous row range with resident
    weights; no data-path collective (one all-gather of row counts at the end).
"""
import torch
import torch.distributed as dist


def shard_range(n, world, rank):
    """Rows [lo, hi) of shard `rank` (contiguous, ceil(N/G) rows per shard)."""
    per = -(-n // world) if n > 0 else 0
    lo = min(n, rank * per)
    hi = min(n, lo + per)
    return lo, hi


class ShardedIndexFlatIP:
    """The IndexFlatIP surface over a corpus sharded by rows across the ranks of a process group.

    index_factory(d, id_base) -> object with add(x)/search(q, k)/ntotal (default: the HIP index on
    this rank's GPU); merge_fn(Dc, Ic) -> (D, I) merges [G, nq, k] candidates (default: the GPU
    merge kernel).  Both are injectable so that the collective protocol is testable on CPU."""

    # candidate bytes over all ranks from which the exchange is sliced by queries ("auto")
    SLICED_FROM_BYTES = 64 << 20

    def __init__(self, d, group=None, device=None, index_factory=None, merge_fn=None, exchange="auto"):
        if exchange not in ("auto", "gather", "sliced"):
            raise ValueError("exchange must be 'auto', 'gather' or 'sliced'")
        self.exchange = exchange
        self.d = int(d)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        if index_factory is None:
            from .index import IndexFlatIP, merge_topk
            dev = torch.cuda.current_device() if device is None else torch.device(device).index
            index_factory = lambda dd, base: IndexFlatIP(dd, device=dev, id_base=base)  # noqa: E731
            merge_fn = merge_fn or (lambda Dc, Ic: merge_topk(Dc, Ic, device=dev))
        if merge_fn is None:
            raise ValueError("merge_fn is required with a custom index_factory")
        self._factory = index_factory
        self._merge = merge_fn
        self.local = None
        self.ntotal = 0
        self.lo = self.hi = 0

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

    def add_shard(self, x_local, id_base, ntotal):
        """This rank's rows are already local (e.g. just written by encode_rows): global ids are
        id_base + row."""
        self.local = self._factory(self.d, int(id_base))
        self.lo, self.hi = int(id_base), int(id_base) + x_local.shape[0]
        if x_local.shape[0]:
            self.local.add(x_local)
        self.ntotal = int(ntotal)

    def search(self, q, k):
        """D, I = search(q, k) on every rank (same result everywhere)."""
        if self.local is None:
            raise RuntimeError("search before add")
        D, I = self.local.search(q, k)
        if self.world == 1:
            return D, I
        if not torch.is_tensor(D):
            D, I = torch.from_numpy(D), torch.from_numpy(I)
        D = D.contiguous()
        I = I.contiguous()
        nq, kk = D.shape
        mode = self.exchange
        if mode == "auto":
            mode = "sliced" if self.world * nq * kk * 12 >= self.SLICED_FROM_BYTES else "gather"
        if mode == "sliced":
            return self._exchange_sliced(D, I)
        # concatenated along dim 0 (the layout every backend's all_gather_into_tensor accepts)
        Dg = torch.empty((self.world * nq, kk), dtype=D.dtype, device=D.device)
        Ig = torch.empty((self.world * nq, kk), dtype=I.dtype, device=I.device)
        # candidate exchange: 12 B x nq x k per rank (2.66 MB at nq = 2214, k = 100)
        dist.all_gather_into_tensor(Dg, D, group=self.group)
        dist.all_gather_into_tensor(Ig, I, group=self.group)
        return self._merge(Dg.view(self.world, nq, kk), Ig.view(self.world, nq, kk))


def _exchange_sliced_impl(self, D, I):
    """SURVEY.md §8e option B.  Rank r owns queries [r*per, (r+1)*per), per = ceil(nq/G); rows past nq
    are padding (score -FLT_MAX, id -1: the "missing" convention the merge already handles)."""
    G = self.world
    nq, kk = D.shape
    per = -(-nq // G)
    pad = per * G
    if pad != nq:
        Dp = torch.full((pad, kk), -3.4028234663852886e38, dtype=D.dtype, device=D.device)
        Ip = torch.full((pad, kk), -1, dtype=I.dtype, device=I.device)
        Dp[:nq] = D
        Ip[:nq] = I
    else:
        Dp, Ip = D, I
    # all-to-all with equal splits: chunk r of my rows goes to rank r; I receive chunk `rank` of every rank
    Dr = torch.empty_like(Dp)
    Ir = torch.empty_like(Ip)
    dist.all_to_all_single(Dr, Dp, group=self.group)
    dist.all_to_all_single(Ir, Ip, group=self.group)
    Dm, Im = self._merge(Dr.view(G, per, kk), Ir.view(G, per, kk))
    if not torch.is_tensor(Dm):
        Dm, Im = torch.from_numpy(Dm), torch.from_numpy(Im)
    Dm = Dm.to(D.device).contiguous()
    Im = Im.to(I.device).contiguous()
    Dg = torch.empty((pad, kk), dtype=D.dtype, device=D.device)
    Ig = torch.empty((pad, kk), dtype=I.dtype, device=I.device)
    dist.all_gather_into_tensor(Dg, Dm, group=self.group)
    dist.all_gather_into_tensor(Ig, Im, group=self.group)
    return Dg[:nq], Ig[:nq]


ShardedIndexFlatIP._exchange_sliced = _exchange_sliced_impl


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
