"""read_index / write_index for flat inner-product indexes (scripts/eval_ottqa_retrieval.py:106-107,114-115:
`faiss.read_index(args.faiss_save_path)` when the file exists; the matching `faiss.write_index` is commented
out there).

The file layout is faiss' own for IndexFlatIP, restated from its published serialiser
(faiss/impl/index_write.cpp: write_index_header + WRITE_XBVECTOR) -- faiss is not installed here and no
.index file ships with the reference, so this reader/writer is **parity unpinned** against real faiss files;
it validates everything it can (magic, metric, sizes against the file length) and refuses anything else:

    uint32  fourcc            'IxFI' (flat, inner product)
    int32   d
    int64   ntotal
    int64   dummy, int64 dummy            (1 << 20 each)
    uint8   is_trained
    int32   metric_type                   0 = METRIC_INNER_PRODUCT
    uint64  size                          number of floats (faiss >= 1.7.1, byte codes) or floats / 4 (<= 1.7.0)
    float32 xb[ntotal * d]                row-major

HNSW files ('IHNf' ...) are out of scope (north_star: exact search) and raise.
"""
import os
import struct

import numpy as np

_FOURCC_IP = int.from_bytes(b"IxFI", "little")
_HEADER = struct.Struct("<IiqqqBi")          # fourcc, d, ntotal, dummy, dummy, is_trained, metric_type


def write_index(index, path):
    """Serialise a flat IP index (any object with d, ntotal, reconstruct_n) in faiss' IndexFlatIP layout."""
    d, n = int(index.d), int(index.ntotal)
    tmp = path + ".tmp"
    with open(tmp, "wb") as f:
        f.write(_HEADER.pack(_FOURCC_IP, d, n, 1 << 20, 1 << 20, 1, 0))
        f.write(struct.pack("<Q", n * d))
        step = 1 << 16
        for lo in range(0, n, step):
            rows = index.reconstruct_n(lo, min(step, n - lo))
            f.write(np.ascontiguousarray(rows, dtype="<f4").tobytes())
    os.replace(tmp, path)


def read_flat(path):
    """(d, memmap float32 [ntotal, d]) of a faiss IndexFlatIP file; ValueError on anything else."""
    size = os.path.getsize(path)
    with open(path, "rb") as f:
        head = f.read(_HEADER.size + 8)
    if len(head) < _HEADER.size + 8:
        raise ValueError("%s: too short for an index header" % path)
    fourcc, d, n, _, _, _trained, metric = _HEADER.unpack_from(head)
    if fourcc != _FOURCC_IP:
        raise ValueError("%s: fourcc %r is not a flat inner-product index ('IxFI'); HNSW and other types are not built"
                         % (path, head[:4]))
    if metric != 0 or d <= 0 or n < 0:
        raise ValueError("%s: unexpected header (d=%d ntotal=%d metric=%d)" % (path, d, n, metric))
    (cnt,) = struct.unpack_from("<Q", head, _HEADER.size)
    if cnt not in (n * d, (n * d) // 4) or (cnt != n * d and (n * d) % 4):
        raise ValueError("%s: vector count %d does not match d*ntotal = %d" % (path, cnt, n * d))
    off = _HEADER.size + 8
    if size != off + n * d * 4:
        raise ValueError("%s: %d bytes, expected %d" % (path, size, off + n * d * 4))
    xb = np.memmap(path, dtype="<f4", mode="r", offset=off, shape=(n, d)) if n else np.zeros((0, d), np.float32)
    return d, xb


def read_index(path, device=0, dtype="fp32"):
    """faiss.read_index for flat IP files: an IndexFlatIP on the GPU holding the file's rows."""
    from .index import IndexFlatIP
    d, xb = read_flat(path)
    index = IndexFlatIP(d, device=device, dtype=dtype)
    step = 1 << 18
    for lo in range(0, xb.shape[0], step):
        index.add(np.ascontiguousarray(xb[lo:lo + step], dtype=np.float32))
    return index
