"""tools/stress_poll.py [--calls N] — thousands of consecutive searches on ONE index, batch sizes hopping between the
small-batch path (finish kernel, per-query tickets), the 33 ... 64-query stream kernel and the large-batch path
(optimistic threshold, margin check's ticket), every pass ending in a polled proof word: every result must equal the
first result of its batch size, which is checked against the oracle once.  A stale poll word, a ticket left non-zero or
a counter shared between two passes shows up as a wrong or hanging call."""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import flat_ip  # noqa: E402
from otter_amd.index import IndexFlatIP  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--calls", type=int, default=4000)
ap.add_argument("--rows", type=int, default=300000)
args = ap.parse_args()
rng = np.random.RandomState(5)
n, d, k = args.rows, 768, 100
xb = rng.randn(n, d).astype(np.float32)
q = rng.randn(700, d).astype(np.float32)
ix = IndexFlatIP(d)
ix.add(xb)
sizes = [1, 2, 8, 16, 31, 32, 33, 48, 64, 65, 130, 256, 257, 700]
ref = {}
for nq in sizes:
    D, I = ix.search(q[:nq], k)
    Dr, Ir = flat_ip.search_c(xb, q[:nq], k, mode=flat_ip.MODE_GPU_ORDER)
    assert np.array_equal(I, Ir) and np.array_equal(D.view(np.uint32), Dr.view(np.uint32)), nq
    ref[nq] = (D, I)
print("references ok", flush=True)
t0 = time.time()
bad = 0
for c in range(args.calls):
    nq = sizes[int(rng.randint(0, len(sizes)))]
    D, I = ix.search(q[:nq], k)
    if not (np.array_equal(I, ref[nq][1]) and np.array_equal(D.view(np.uint32), ref[nq][0].view(np.uint32))):
        bad += 1
        print("call %d nq %d differs" % (c, nq), flush=True)
    if c % 500 == 499:
        print("%d calls, %d different, %.1f s" % (c + 1, bad, time.time() - t0), flush=True)
print("calls: %d, different: %d, refined %d fallback %d" % (args.calls, bad, ix.last_refined, ix.last_fallback))
sys.exit(1 if bad else 0)
