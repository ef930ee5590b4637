import torch, time, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from otter_amd.index import merge_topk
for G in (2, 4, 8):
    D = torch.randn((G, 2214, 100), device='cuda').sort(dim=2, descending=True).values.contiguous()
    I = torch.randint(0, 1 << 40, (G, 2214, 100), device='cuda')
    for _ in range(3): merge_topk(D, I, device=0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): merge_topk(D, I, device=0)
    torch.cuda.synchronize(); print(G, (time.perf_counter() - t0) / 20 * 1e3, 'ms')
