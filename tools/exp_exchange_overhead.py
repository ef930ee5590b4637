"""tools/exp_exchange_overhead.py — what ShardedIndexFlatIP's exchange adds around the collective on one rank:
pack (scores + shard-local ids into the send buffer), unpack (ids back to global) and the G-way merge, timed on one GPU
with a fabricated [G, nq, k] receive buffer (no process group: the all-gather itself is not in this figure)."""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otter_amd.sharded import PackedCandidates, to_local_ids, to_global_ids
from otter_amd.index import merge_topk

G, nq, k = 8, 2214, 100
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
D = torch.sort(torch.randn((nq, k), device=dev, generator=g), dim=1, descending=True)[0].contiguous()
I = torch.randint(0, 1000000, (nq, k), device=dev, generator=g) + 3000000
bases = (torch.arange(G) * 1000000).to(torch.int64)
bases_dev = bases.to(dev)


def once(id_bytes):
    send = PackedCandidates(nq, k, dev, id_bytes=id_bytes)
    Ds, Is = send.views()
    Ds.copy_(D)
    Is.copy_(to_local_ids(I, 3000000) if id_bytes == 4 else I)
    recv = send.buf.repeat(G)                       # stands in for the all-gather
    Dg, Ig = send.views(recv.view(G, send.nbytes))
    if id_bytes == 4:
        Ig = to_global_ids(Ig, bases_dev)
    return merge_topk(Dg.contiguous(), Ig.contiguous(), device=0)


for idb in (4, 8):
    for _ in range(5):
        once(idb)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        once(idb)
    torch.cuda.synchronize()
    print("id_bytes %d: %.1f us per search around the collective (G=%d, nq=%d, k=%d)" % (idb, (time.perf_counter() - t0) / 50 * 1e6, G, nq, k))
