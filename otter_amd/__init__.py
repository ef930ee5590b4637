"""otter_amd — MI355X-native (gfx950) dense-retrieval hot path of OTTeR.

Two drop-in surfaces over one C-ABI library (include/otter_hip.h, libotter_hip.so):
  * otter_amd.index.IndexFlatIP      — faiss.IndexFlatIP add()/search() (reference
    scripts/eval_ottqa_retrieval.py:108-109,129)
  * otter_amd.encoder                — the bi-encoder forward (reference
    retrieval/models/tb_retriever.py:150-261, retrieval/models/retriever.py:249-345)
"""
__version__ = "0.1.0"
