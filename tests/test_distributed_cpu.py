"""world_size-2 gloo tests (CPU) of the multi-GPU host logic: row sharding, the candidate
all-gather + merge protocol of ShardedIndexFlatIP, and the data-parallel encode partition.
The per-shard search and the merge are injected from the oracle here (tests may use it); on the
GPU box the same class runs with the HIP index and the GPU merge kernel."""
import os
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import flat_ip
from otter_amd import sharded
from otter_amd.encode_driver import collate_tokens, em_collate_bert, predict


class _OracleIndex:
    def __init__(self, d, id_base):
        self.ix = flat_ip.IndexFlatIP(d)
        self.base = id_base

    def add(self, x):
        self.ix.add(np.asarray(x))

    @property
    def ntotal(self):
        return self.ix.ntotal

    def search(self, q, k):
        D, I = self.ix.search(np.asarray(q), k)
        I = np.where(I >= 0, I + self.base, -1)
        return torch.from_numpy(D), torch.from_numpy(I)


def _merge(Dc, Ic):
    D, I = flat_ip.merge_c(Dc.numpy(), Ic.numpy())
    return torch.from_numpy(D), torch.from_numpy(I)


def _worker(rank, world, initfile, n, outdir, exchange="auto", compact=True, via_add_shard=False):
    dist.init_process_group("gloo", init_method="file://" + initfile, rank=rank, world_size=world)
    try:
        rng = np.random.RandomState(7)
        xb = rng.randn(n, 64).astype(np.float32)
        xb[n // 2 + 3] = xb[5]                       # a cross-shard exact tie
        q = rng.randn(9, 64).astype(np.float32)
        k = 20
        six = sharded.ShardedIndexFlatIP(64, index_factory=_OracleIndex, merge_fn=_merge, exchange=exchange,
                                         compact=compact)
        if via_add_shard:
            lo, hi = sharded.shard_range(n, world, rank)
            six.add_shard(xb[lo:hi], lo, n)          # every rank learns the others' bases by one all-gather
        else:
            six.add_global(xb)
        assert six.ntotal == n and (six.lo, six.hi) == sharded.shard_range(n, world, rank)
        assert six.bases.tolist() == [sharded.shard_range(n, world, r)[0] for r in range(world)]
        D, I = six.search(q, k)
        if exchange == "gather":                     # 8 bytes per candidate on the wire (12 with int64 ids)
            per = 8 if compact else 12
            assert six.last_exchange_bytes == (9 * k * per + 15) // 16 * 16
        np.save(os.path.join(outdir, "D%d.npy" % rank), D.numpy())
        np.save(os.path.join(outdir, "I%d.npy" % rank), I.numpy())

        # data-parallel encode: a fake model whose embedding encodes the row id
        def batches(lo, hi, bs=4):
            for s in range(lo, hi, bs):
                ids = torch.arange(s, min(hi, s + bs)).view(-1, 1)
                yield {"input_ids": ids}

        model = lambda b: {"embed": b["input_ids"].float() * 2.0}   # noqa: E731
        emb, lo, counts = sharded.encode_rows(model, batches, 11)
        assert sum(counts) == 11 and counts[rank] == emb.shape[0]
        assert torch.equal(emb[:, 0], torch.arange(lo, lo + emb.shape[0]).float() * 2.0)
    finally:
        dist.destroy_process_group()


# 7 rows < k: -1 padding survives the merge.  "sliced" = the all-to-all exchange by query slices
# (9 queries over 2 or 3 ranks: ragged last slice, padded rows must not leak into the result)
@pytest.mark.parametrize("n,world,exchange,compact,via_add_shard", [
    (101, 2, "gather", True, False), (7, 2, "gather", True, False), (101, 2, "sliced", True, False),
    (7, 2, "sliced", True, False), (101, 3, "sliced", True, False), (101, 3, "gather", True, True),
    (101, 2, "gather", False, False), (101, 3, "sliced", False, True)])
def test_sharded_search_gloo(n, world, exchange, compact, via_add_shard):
    with tempfile.TemporaryDirectory() as tmp:
        initfile = os.path.join(tmp, "init")
        mp.spawn(_worker, args=(world, initfile, n, tmp, exchange, compact, via_add_shard), nprocs=world, join=True)
        rng = np.random.RandomState(7)
        xb = rng.randn(n, 64).astype(np.float32)
        xb[n // 2 + 3] = xb[5]
        q = rng.randn(9, 64).astype(np.float32)
        ref = flat_ip.IndexFlatIP(64)
        ref.add(xb)
        Dr, Ir = ref.search(q, 20)
        for r in range(world):
            np.testing.assert_array_equal(np.load(os.path.join(tmp, "I%d.npy" % r)), Ir)
            np.testing.assert_allclose(np.load(os.path.join(tmp, "D%d.npy" % r)), Dr, rtol=0, atol=1e-5)


def test_shard_range_covers_everything():
    for n in (0, 1, 7, 8, 9, 1000003):
        for world in (1, 2, 3, 8):
            spans = [sharded.shard_range(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert all(lo <= hi for lo, hi in spans)


def test_collate_and_predict_never_drop_rows():
    items = [{"tb": {"input_ids": torch.tensor([[0, 5, 2]]), "attention_mask": torch.tensor([[1, 1, 1]]),
                     "part2_mask": torch.tensor([[0, 1, 0]]), "part3_mask": torch.tensor([[0, 1, 0]])}},
             {"tb": {"input_ids": torch.tensor([[0, 5, 6, 7, 2]]), "attention_mask": torch.ones(1, 5, dtype=torch.long),
                     "part2_mask": torch.tensor([[0, 1, 1, 1, 0]]), "part3_mask": torch.tensor([[0, 1, 1, 1, 0]])}}]
    b = em_collate_bert(items, pad_id=1)
    assert b["input_ids"].tolist() == [[0, 5, 2, 1, 1], [0, 5, 6, 7, 2]]
    assert b["input_mask"].tolist() == [[1, 1, 1, 0, 0], [1, 1, 1, 1, 1]]
    assert set(b) == {"input_ids", "input_mask", "part2_mask", "part3_mask"}
    assert collate_tokens([torch.tensor([1]), torch.tensor([1, 2])], 9).tolist() == [[1, 9], [1, 2]]

    class Boom:
        def eval(self):
            return self

        def __call__(self, batch):
            raise RuntimeError("kernel failed")

    import otter_amd.encode_driver as drv
    old = drv.move_to_cuda
    drv.move_to_cuda = lambda x: x
    try:
        with pytest.raises(RuntimeError):
            predict(Boom(), [b])
    finally:
        drv.move_to_cuda = old


def test_predict_pipeline_keeps_order_and_row_counts():
    """Two batches in flight, slots reused: ragged batch sizes and sequence lengths, every row accounted for
    in dataloader order (the host logic of the pipelined predict(); the stream overlap runs on the GPU box)."""
    class Model:
        def __init__(self):
            self.seen = []

        def eval(self):
            return self

        def __call__(self, batch):
            ids = batch["input_ids"]
            self.seen.append(tuple(ids.shape))
            # embedding = (first token id, sum of the mask) per row
            return {"embed": torch.stack([ids[:, 0].float(), batch["input_mask"].sum(1).float()], dim=1)}

    batches = []
    row = 0
    for bs, L in [(4, 7), (4, 3), (1, 9), (3, 9), (2, 1)]:
        ids = torch.arange(row, row + bs).view(-1, 1).repeat(1, L)
        mask = torch.ones((bs, L), dtype=torch.int32)      # a non-int64 tensor is converted, not rejected
        batches.append({"input_ids": ids, "input_mask": mask})
        row += bs
    m = Model()
    emb = predict(m, iter(batches))
    assert emb.shape == (row, 2)
    assert torch.equal(emb[:, 0], torch.arange(row).float())
    assert emb[:, 1].tolist() == [7.0] * 4 + [3.0] * 4 + [9.0] * 4 + [1.0] * 2
    assert m.seen == [(4, 7), (4, 3), (1, 9), (3, 9), (2, 1)]
    with pytest.raises(ValueError):
        predict(m, iter([]))


def test_sharded_search_with_numpy_queries_and_a_numpy_index():
    """a factory whose search returns numpy (as the HIP index does for host queries) must work
    at world > 1 -- single process here, world 1 falls through; the packing helpers are checked directly."""
    pc = sharded.PackedCandidates(3, 5, "cpu")
    D, I = pc.views()
    D.copy_(torch.arange(15, dtype=torch.float32).view(3, 5))
    I.copy_(torch.arange(15, dtype=torch.int64).view(3, 5) * 7)
    two = torch.cat([pc.buf, pc.buf]).view(2, pc.nbytes)
    D2, I2 = pc.views(two)
    assert D2.shape == (2, 3, 5) and torch.equal(D2[1], D) and torch.equal(I2[0], I)
    assert pc.nbytes % 16 == 0 and pc.nbytes >= 15 * 12


def test_compact_ids_round_trip():
    I = torch.tensor([[5, 7, -1], [100, -1, -1]], dtype=torch.int64)
    loc = sharded.to_local_ids(I, 5)
    assert loc.dtype == torch.int32 and loc.tolist() == [[0, 2, -1], [95, -1, -1]]
    both = torch.stack([loc, loc])
    glob = sharded.to_global_ids(both, torch.tensor([5, 1000]))
    assert glob.dtype == torch.int64 and glob[0].tolist() == I.tolist() and glob[1].tolist() == [[1000, 1002, -1], [1095, -1, -1]]
    pc = sharded.PackedCandidates(2, 3, "cpu", id_bytes=4)
    D, Iv = pc.views()
    assert Iv.dtype == torch.int32 and pc.nbytes == (2 * 3 * 8 + 15) // 16 * 16
