"""IndexFlatIP — the faiss surface the reference's eval path uses, on one MI355X.

Mirrors `index = faiss.IndexFlatIP(d); index.add(xb); D, I = index.search(q, k)`
(reference scripts/eval_ottqa_retrieval.py:108-109,129): same names, same argument
meaning, same result layout (D float32 [nq, k] descending, I int64 [nq, k], -1 padding).
numpy in -> numpy out (host copies); torch device tensors in -> torch device tensors
out (zero copy; work is enqueued on the current stream -- a search of an fp32 index of 65,536 and
more rows also waits for it once per pass of its coarse search, see include/otter_hip.h).

All compute happens in libotter_hip.so (otter_amd/csrc/index.hip); there is no CPU path.
"""
import ctypes

import numpy as np

from . import _lib

try:  # torch is plumbing only (device memory + streams); numpy-only callers do not need it
    import torch
except Exception:  # pragma: no cover
    torch = None


def _is_torch(x):
    return torch is not None and isinstance(x, torch.Tensor)


def _stream_ptr(device_index):
    if torch is None or not torch.cuda.is_available():
        return None
    return ctypes.c_void_p(torch.cuda.current_stream(device_index).cuda_stream)


class IndexFlatIP:
    """Exact inner-product index resident in the HBM of one GPU.

    Parameters
    ----------
    d : int            vector dimension (768, or 768*3 for the reference's three-cat embeddings)
    device : int       HIP device ordinal
    id_base : int      added to every reported id (row-sharded use: the shard's first global row)
    dtype : str        'fp32' = IndexFlatIP's arithmetic; 'bf16' = rows and queries rounded to bfloat16
    planes : str       HBM economy of an fp32 index: 'all' (default; 2.5x the corpus' bytes once the first search has
                       built the bf16 planes of the coarse pass), 'hi' (1.5x: the one-product pass only, what it cannot
                       prove goes to the exact kernels -- the choice for serving small batches), 'none' (1.0x: exact
                       kernels only).  Same result bits under every setting (otter_index_set_planes).
    """

    PLANES = {"none": 0, "hi": 1, "all": 2}

    def __init__(self, d, device=0, id_base=0, dtype="fp32", planes="all"):
        self._lib = _lib.load()
        self.d = int(d)
        self.device = int(device)
        if dtype not in ("fp32", "bf16"):
            raise ValueError("dtype must be 'fp32' (IndexFlatIP arithmetic) or 'bf16' (storage of the 100M-row config)")
        self.dtype = dtype
        self._h = ctypes.c_void_p()
        _lib.check(self._lib.otter_index_create(self.d, _lib.OTTER_BF16 if dtype == "bf16" else _lib.OTTER_F32,
                                                self.device, ctypes.byref(self._h)), "otter_index_create")
        self.id_base = 0
        if id_base:
            self.set_id_base(id_base)
        self.planes = "all"
        if planes != "all":
            self.set_planes(planes)

    # -- faiss attributes
    @property
    def ntotal(self):
        return int(self._lib.otter_index_ntotal(self._h))

    @property
    def is_trained(self):
        return True

    def set_planes(self, planes):
        """'none' | 'hi' | 'all': which bf16 planes the index keeps beside its fp32 rows (see the class docstring)."""
        if planes not in self.PLANES:
            raise ValueError("planes must be 'none', 'hi' or 'all'")
        _lib.check(self._lib.otter_index_set_planes(self._h, self.PLANES[planes]), "otter_index_set_planes")
        self.planes = planes

    def set_id_base(self, id_base):
        _lib.check(self._lib.otter_index_set_id_base(self._h, int(id_base)), "otter_index_set_id_base")
        self.id_base = int(id_base)

    def add(self, x):
        """index.add(xb): x float32 [n, d], numpy (host) or torch tensor on this index's device."""
        if _is_torch(x):
            if x.dim() != 2 or x.shape[1] != self.d:
                raise ValueError("add: expected [n, %d], got %r" % (self.d, tuple(x.shape)))
            if x.is_cuda:
                if x.device.index != self.device:
                    raise ValueError("add: tensor on %s, index on device %d" % (x.device, self.device))
                x = x.to(torch.float32).contiguous()
                _lib.check(self._lib.otter_index_add(self._h, ctypes.c_void_p(x.data_ptr()), x.shape[0], 1,
                                                     _stream_ptr(self.device)), "otter_index_add")
                # the copy into the index's private HBM is only enqueued: keep x alive until it ran
                torch.cuda.current_stream(self.device).synchronize()
                return
            x = x.detach().cpu().numpy()
        x = np.ascontiguousarray(x, dtype=np.float32)
        if x.ndim != 2 or x.shape[1] != self.d:
            raise ValueError("add: expected [n, %d] float32, got %r" % (self.d, x.shape))
        _lib.check(self._lib.otter_index_add(self._h, ctypes.c_void_p(x.ctypes.data), x.shape[0], 0,
                                             _stream_ptr(self.device)), "otter_index_add")

    def search(self, q, k, out=None):
        """D, I = index.search(q, k).  out=(D, I): device queries only -- contiguous float32 / int64 [nq, k]
        tensors on this index's device that take the result (e.g. views of an exchange buffer)."""
        k = int(k)
        if k < 1 or k > _lib.OTTER_MAX_K:
            raise ValueError("search: k=%d outside [1, %d]" % (k, _lib.OTTER_MAX_K))
        if _is_torch(q) and q.is_cuda:
            if q.dim() != 2 or q.shape[1] != self.d:
                raise ValueError("search: expected [nq, %d], got %r" % (self.d, tuple(q.shape)))
            if q.device.index != self.device:
                raise ValueError("search: tensor on %s, index on device %d" % (q.device, self.device))
            q = q.to(torch.float32).contiguous()
            nq = q.shape[0]
            if out is not None:
                D, I = out
                for t, dt in ((D, torch.float32), (I, torch.int64)):
                    if not (_is_torch(t) and t.is_cuda and t.device == q.device and t.dtype == dt and
                            tuple(t.shape) == (nq, k) and t.is_contiguous()):
                        raise ValueError("search: out must be contiguous (float32, int64) [%d, %d] tensors on %s"
                                         % (nq, k, q.device))
            else:
                D = torch.empty((nq, k), dtype=torch.float32, device=q.device)
                I = torch.empty((nq, k), dtype=torch.int64, device=q.device)
            _lib.check(self._lib.otter_index_search(self._h, ctypes.c_void_p(q.data_ptr()), nq, k,
                                                    ctypes.c_void_p(D.data_ptr()), ctypes.c_void_p(I.data_ptr()), 1,
                                                    _stream_ptr(self.device)), "otter_index_search")
            return D, I
        if out is not None:
            raise ValueError("search: out= is for queries on the index's device")
        if _is_torch(q):
            q = q.detach().numpy()
        q = np.ascontiguousarray(q, dtype=np.float32)
        if q.ndim != 2 or q.shape[1] != self.d:
            raise ValueError("search: expected [nq, %d] float32, got %r" % (self.d, q.shape))
        nq = q.shape[0]
        D = np.empty((nq, k), dtype=np.float32)
        I = np.empty((nq, k), dtype=np.int64)
        _lib.check(self._lib.otter_index_search(self._h, ctypes.c_void_p(q.ctypes.data), nq, k,
                                                ctypes.c_void_p(D.ctypes.data), ctypes.c_void_p(I.ctypes.data), 0,
                                                _stream_ptr(self.device)), "otter_index_search")
        return D, I

    def search_noalloc(self, q, k, D, I):
        """otter_index_search_noalloc: at most 32 queries, k <= 128, fp32 index; q, D, I are caller-owned tensors
        on this index's device (float32 [nq, d], float32 [nq, k], int64 [nq, k], contiguous).  The library call
        allocates nothing (its scratch is a static per-GPU array); work is only enqueued on the current stream."""
        for t, dt, shape in ((q, torch.float32, (q.shape[0], self.d)), (D, torch.float32, (q.shape[0], int(k))),
                             (I, torch.int64, (q.shape[0], int(k)))):
            if not (_is_torch(t) and t.is_cuda and t.device.index == self.device and t.dtype == dt and
                    tuple(t.shape) == shape and t.is_contiguous()):
                raise ValueError("search_noalloc: expected a contiguous %s %r tensor on device %d" % (dt, shape, self.device))
        _lib.check(self._lib.otter_index_search_noalloc(self._h, ctypes.c_void_p(q.data_ptr()), q.shape[0], int(k),
                                                        ctypes.c_void_p(D.data_ptr()), ctypes.c_void_p(I.data_ptr()),
                                                        _stream_ptr(self.device)), "otter_index_search_noalloc")
        return D, I

    def search_subsets(self, q, k, subsets, return_positions=False):
        """Per-query search restricted to row subsets: what the reference does by building a throw-away
        faiss.IndexFlatIP over each question's TF-IDF-filtered blocks
        (retrieval/eval_ottqa_retrieval_tfidffilter.py:530-546), as one batched call on the resident corpus.

        subsets: a sequence of nq 1-D integer arrays of row ids (ragged), or an int64 [nq, m] array / CUDA
        tensor padded with -1.  Returns (D, I): I holds row ids, or -- return_positions=True, the
        reference's convention (its small index numbers the subset's rows 0..len-1) -- positions in the
        subset (the first position of a repeated id)."""
        k = int(k)
        if k < 1 or k > _lib.OTTER_MAX_K:
            raise ValueError("search_subsets: k=%d outside [1, %d]" % (k, _lib.OTTER_MAX_K))
        on_dev = _is_torch(q) and q.is_cuda
        if _is_torch(subsets):
            ids = subsets.to(torch.int64)
        elif isinstance(subsets, np.ndarray) and subsets.ndim == 2:
            ids = np.ascontiguousarray(subsets, dtype=np.int64)
        else:
            lens = [len(s) for s in subsets]
            ids = np.full((len(lens), max(1, max(lens) if lens else 1)), -1, dtype=np.int64)
            for i, s_ in enumerate(subsets):
                ids[i, :lens[i]] = np.asarray(s_, dtype=np.int64)
        nq = q.shape[0]
        if ids.ndim != 2 or ids.shape[0] != nq or ids.shape[1] < 1:
            raise ValueError("search_subsets: need one id list per query, got %r for %d queries" % (tuple(ids.shape), nq))
        m = int(ids.shape[1])
        if on_dev:
            if q.dim() != 2 or q.shape[1] != self.d or q.device.index != self.device:
                raise ValueError("search_subsets: expected [nq, %d] on device %d" % (self.d, self.device))
            q = q.to(torch.float32).contiguous()
            ids_d = (ids if _is_torch(ids) else torch.from_numpy(ids)).to(q.device).contiguous()
            D = torch.empty((nq, k), dtype=torch.float32, device=q.device)
            I = torch.empty((nq, k), dtype=torch.int64, device=q.device)
            if nq:
                _lib.check(self._lib.otter_index_search_subset(
                    self._h, ctypes.c_void_p(q.data_ptr()), nq, k, ctypes.c_void_p(ids_d.data_ptr()), m,
                    ctypes.c_void_p(D.data_ptr()), ctypes.c_void_p(I.data_ptr()), 1, _stream_ptr(self.device)),
                    "otter_index_search_subset")
            if return_positions:
                eq = I.unsqueeze(2) == ids_d.unsqueeze(1)                       # [nq, k, m]
                pos = torch.where(eq.any(2) & (I >= 0), eq.to(torch.int8).argmax(2), torch.full_like(I, -1))
                return D, pos
            return D, I
        if _is_torch(q):
            q = q.detach().numpy()
        q = np.ascontiguousarray(q, dtype=np.float32)
        if q.ndim != 2 or q.shape[1] != self.d:
            raise ValueError("search_subsets: expected [nq, %d] float32, got %r" % (self.d, q.shape))
        if _is_torch(ids):
            ids = ids.cpu().numpy()
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        D = np.empty((nq, k), dtype=np.float32)
        I = np.empty((nq, k), dtype=np.int64)
        if nq:
            _lib.check(self._lib.otter_index_search_subset(
                self._h, ctypes.c_void_p(q.ctypes.data), nq, k, ctypes.c_void_p(ids.ctypes.data), m,
                ctypes.c_void_p(D.ctypes.data), ctypes.c_void_p(I.ctypes.data), 0, _stream_ptr(self.device)),
                "otter_index_search_subset")
        if return_positions:
            pos = np.full_like(I, -1)
            for i in range(nq):
                first = {}
                for j, v in enumerate(ids[i]):
                    if v >= 0 and int(v) not in first:
                        first[int(v)] = j
                pos[i] = [first.get(int(v), -1) if v >= 0 else -1 for v in I[i]]
            return D, pos
        return D, I

    def reconstruct_n(self, i0=0, n=None):
        """float32 [n, d] copy of rows [i0, i0 + n) (faiss: index.reconstruct_n)."""
        n = self.ntotal - i0 if n is None else int(n)
        out = np.empty((n, self.d), dtype=np.float32)
        _lib.check(self._lib.otter_index_reconstruct_n(self._h, int(i0), n, ctypes.c_void_p(out.ctypes.data), 0,
                                                       _stream_ptr(self.device)), "otter_index_reconstruct_n")
        return out

    @property
    def coarse_state(self):
        """0: bf16 planes of the coarse pass not built yet, 1: in use, 2: no room in HBM (the exact kernel serves
        every batch size), 3: not an fp32 index, 4: switched off for this corpus until reset() (its proofs failed for
        most queries: scores closer together than bf16 products resolve)."""
        return int(self._lib.otter_index_coarse_state(self._h))

    @property
    def last_fallback(self):
        """Queries of the last search() that the coarse bf16 pass could not prove and the exact kernel re-did."""
        return int(self._lib.otter_index_last_fallback(self._h))

    @property
    def last_refined(self):
        """Queries of the last search() that the two-product coarse pass left unproven (searched again)."""
        return int(self._lib.otter_index_last_refined(self._h))

    def reset(self):
        _lib.check(self._lib.otter_index_reset(self._h), "otter_index_reset")

    # -- measurement hook (bench.py): HIP-event time of the scan kernel
    def set_timing(self, enable=True):
        _lib.check(self._lib.otter_index_set_timing(self._h, 1 if enable else 0), "otter_index_set_timing")

    def timing(self):
        """(summed scan-kernel milliseconds, launches) since the previous call."""
        ms = ctypes.c_double(0.0)
        n = ctypes.c_int64(0)
        _lib.check(self._lib.otter_index_timing(self._h, ctypes.byref(ms), ctypes.byref(n)), "otter_index_timing")
        return ms.value, n.value

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.otter_index_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MultiGpuIndexFlatIP:
    """One process, every GPU: what `index = faiss.index_cpu_to_all_gpus(index)` gives the reference
    (scripts/eval_ottqa_retrieval.py:110-112), with the rows SHARDED over the devices instead of replicated.

    devices: HIP ordinals, one shard each (default: every visible GPU; an ordinal may repeat -- several shards
    on one GPU, which is how the one-GPU tests drive it).  Shard g holds rows [g*ceil(N/G), ...) of the array
    handed to add(); ids are insertion order.  search(): numpy in -> numpy out, or tensors on devices[0] in ->
    tensors there out.  All compute and the exchange are otter_sharded_search (include/otter_hip.h)."""

    def __init__(self, d, devices=None, dtype="fp32"):
        self._lib = _lib.load()
        if devices is None:
            devices = list(range(int(self._lib.otter_device_count())))
        devices = [int(x) for x in devices]
        if not devices:
            raise ValueError("MultiGpuIndexFlatIP: no devices")
        self.d, self.dtype, self.devices = int(d), dtype, devices
        self.shards = [IndexFlatIP(d, device=dv, dtype=dtype) for dv in devices]
        self._handles = (ctypes.c_void_p * len(self.shards))(*[s._h for s in self.shards])
        self._added = False

    @property
    def ntotal(self):
        return sum(s.ntotal for s in self.shards)

    @property
    def is_trained(self):
        return True

    def add(self, x):
        """index.add(xb), once: rows are cut into len(devices) contiguous ranges (numpy / memmap / CPU tensor, or a
        tensor on any GPU -- each range is copied to its shard's device)."""
        if self._added:
            raise RuntimeError("MultiGpuIndexFlatIP.add: rows are sharded by contiguous ranges; add once (or reset() first)")
        n, G = int(x.shape[0]), len(self.shards)
        per = -(-n // G) if n else 0
        for g, sh in enumerate(self.shards):
            lo, hi = min(n, g * per), min(n, (g + 1) * per)
            sh.set_id_base(lo)
            for a in range(lo, hi, 1 << 18):          # slab by slab: a memmapped .npy never has to fit in host memory
                part = x[a:min(hi, a + (1 << 18))]
                if _is_torch(part) and part.is_cuda and part.device.index != sh.device:
                    part = part.to("cuda:%d" % sh.device)
                sh.add(part)
        self._added = True

    def reset(self):
        for sh in self.shards:
            sh.reset()
        self._added = False

    def search(self, q, k):
        k = int(k)
        if k < 1 or k > _lib.OTTER_MAX_K:
            raise ValueError("search: k=%d outside [1, %d]" % (k, _lib.OTTER_MAX_K))
        if self._handles is None:
            raise RuntimeError("MultiGpuIndexFlatIP.search: the index was closed")
        home = self.devices[0]
        if _is_torch(q) and q.is_cuda:
            if q.dim() != 2 or q.shape[1] != self.d or q.device.index != home:
                raise ValueError("search: expected [nq, %d] on device %d, got %r on %s" % (self.d, home, tuple(q.shape), q.device))
            q = q.to(torch.float32).contiguous()
            nq = q.shape[0]
            D = torch.empty((nq, k), dtype=torch.float32, device=q.device)
            I = torch.empty((nq, k), dtype=torch.int64, device=q.device)
            _lib.check(self._lib.otter_sharded_search(self._handles, len(self.shards), ctypes.c_void_p(q.data_ptr()), nq, k,
                                                      ctypes.c_void_p(D.data_ptr()), ctypes.c_void_p(I.data_ptr()), 1,
                                                      _stream_ptr(home)), "otter_sharded_search")
            return D, I
        if _is_torch(q):
            q = q.detach().numpy()
        q = np.ascontiguousarray(q, dtype=np.float32)
        if q.ndim != 2 or q.shape[1] != self.d:
            raise ValueError("search: expected [nq, %d] float32, got %r" % (self.d, q.shape))
        nq = q.shape[0]
        D = np.empty((nq, k), dtype=np.float32)
        I = np.empty((nq, k), dtype=np.int64)
        _lib.check(self._lib.otter_sharded_search(self._handles, len(self.shards), ctypes.c_void_p(q.ctypes.data), nq, k,
                                                  ctypes.c_void_p(D.ctypes.data), ctypes.c_void_p(I.ctypes.data), 0,
                                                  _stream_ptr(home)), "otter_sharded_search")
        return D, I

    def close(self):
        self._handles = None          # (the array holds copies of the shards' handles)
        for sh in self.shards:
            sh.close()


def index_cpu_to_all_gpus(index, devices=None):
    """faiss.index_cpu_to_all_gpus(index) (scripts/eval_ottqa_retrieval.py:112): spread a filled IndexFlatIP over
    `devices` (default: every visible GPU) as a MultiGpuIndexFlatIP; the rows leave the source index slab by slab
    (reconstruct_n) and the source is closed.  One device: the index is returned as it is."""
    lib = _lib.load()
    if devices is None:
        devices = list(range(int(lib.otter_device_count())))
    if len(devices) == 1 and int(devices[0]) == index.device:
        return index
    n, G = index.ntotal, len(devices)
    multi = MultiGpuIndexFlatIP(index.d, devices=devices, dtype=index.dtype)
    per = -(-n // G) if n else 0
    slab = 1 << 18
    for g, sh in enumerate(multi.shards):
        lo, hi = min(n, g * per), min(n, (g + 1) * per)
        sh.set_id_base(lo)
        for a in range(lo, hi, slab):
            sh.add(index.reconstruct_n(a, min(slab, hi - a)))
    multi._added = True
    index.close()
    return multi


def merge_topk(Dc, Ic, device=None):
    """Merge G per-shard results: Dc float32 [G, nq, k], Ic int64 [G, nq, k] (torch, on one
    GPU) -> (D [nq, k], I [nq, k]).  Hierarchical when G*k exceeds one workgroup's select."""
    if not (_is_torch(Dc) and Dc.is_cuda and _is_torch(Ic) and Ic.is_cuda):
        raise ValueError("merge_topk: expects CUDA/HIP torch tensors")
    lib = _lib.load()
    G, nq, k = Dc.shape
    dev = Dc.device.index if device is None else int(device)
    Dc = Dc.to(torch.float32).contiguous()
    Ic = Ic.to(torch.int64).contiguous()
    while True:
        g_step = max(2, _lib.MERGE_MAX // k)
        if g_step * k > _lib.MERGE_MAX:
            raise ValueError("merge_topk: k=%d too large to merge" % k)
        groups = [(g0, min(G, g0 + g_step)) for g0 in range(0, G, g_step)]
        Do = torch.empty((len(groups), nq, k), dtype=torch.float32, device=Dc.device)
        Io = torch.empty((len(groups), nq, k), dtype=torch.int64, device=Dc.device)
        for gi, (g0, g1) in enumerate(groups):
            dsub = Dc[g0:g1].contiguous()
            isub = Ic[g0:g1].contiguous()
            _lib.check(lib.otter_merge_topk(dev, ctypes.c_void_p(dsub.data_ptr()), ctypes.c_void_p(isub.data_ptr()),
                                            g1 - g0, nq, k, ctypes.c_void_p(Do[gi].data_ptr()),
                                            ctypes.c_void_p(Io[gi].data_ptr()), _stream_ptr(dev)),
                       "otter_merge_topk")
        if len(groups) == 1:
            return Do[0], Io[0]
        Dc, Ic, G = Do, Io, len(groups)
