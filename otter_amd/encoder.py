"""Bi-encoder forward on MI355X behind the reference's own class surface.

Mirrors (same names, constructor arguments, batch dict and result dict):
  * RobertaSingleEncoderThreeCatPool / SingleEncoderThreeCatPool
        reference retrieval/models/tb_retriever.py:150-227, :250-261   -> {'embed': [B, 3H]}
  * RobertaSingleEncoder / SingleEncoder
        reference retrieval/models/retriever.py:249-311, :334-345      -> {'embed': [B, H]}
  * load_saved(model, path, exact)        reference retrieval/utils/utils.py:12-28
  * move_to_cuda(sample)                  reference retrieval/utils/utils.py:31-48

The classes are thin hosts: `model(batch)` hands the int64 [B, L] tensors of the batch dict
(layout of em_collate_bert, retrieval/data/encode_datasets.py:393-417) to
otter_encoder_forward in libotter_hip.so and returns a float32 device tensor.  There is no
CPU path: without the library or a GPU the call raises.

Differences from the reference, all deliberate:
  * weights come only from load_state_dict / load_saved (no AutoModel.from_pretrained: the
    build has no network); `config` is any object with the HF config attributes;
  * B = 1 works (the reference's 'first' pooling squeezes the batch dimension away and fails);
  * the BERT-family classes run encode_seq / encode_q according to args.encode_table (the
    reference's forward reads an attribute only its Roberta subclasses set);
  * `args.fp16` selects the bf16 compute path (fp32 accumulation) instead of apex amp O1.
"""
import ctypes
from collections import OrderedDict

import time

import numpy as np
import torch

from . import _lib

_IGNORED_SUFFIXES = ("position_ids", "token_type_ids")


def move_to_cuda(sample):
    """retrieval/utils/utils.py:31-48."""
    if len(sample) == 0:
        return {}

    def _move(x):
        if torch.is_tensor(x):
            return x.cuda()
        if isinstance(x, dict):
            return {k: _move(v) for k, v in x.items()}
        if isinstance(x, list):
            return [_move(v) for v in x]
        return x

    return _move(sample)


def load_saved(model, path, exact=True):
    """retrieval/utils/utils.py:12-28: strip a leading 'module.', keep (exact=False) only the keys
    the model has, load non-strictly.  The checkpoint is a plain state_dict of tensors."""
    state_dict = torch.load(path, map_location="cpu", weights_only=True)

    def strip(k):
        return k[7:] if k.startswith("module.") else k

    have = model.state_dict_keys()
    if exact:
        state_dict = {strip(k): v for k, v in state_dict.items()}
    else:
        state_dict = {strip(k): v for k, v in state_dict.items() if strip(k) in have}
    model.load_state_dict(state_dict, False)
    return model


def _cfg_get(config, name, default=None):
    if isinstance(config, dict):
        return config.get(name, default)
    return getattr(config, name, default)


class _EncoderBase:
    """Common host logic; subclasses fix three_cat and the position/type convention."""

    three_cat = True
    roberta = False
    att = False            # ...ThreeCatAtt: attention pooling with its own self_proj_tab / self_proj_psg tensors
    enc_prefix = "encoder"  # attribute name of the HF model in the reference class = its state_dict key prefix
    force_no_proj = None   # the ...CtxEncoder classes fix the projection instead of reading args.no_proj

    def __init__(self, config, args, device=None, dtype=None):
        self.config = config
        self.args = args
        self.shared_encoder = bool(getattr(args, "shared_encoder", True))
        self.no_proj = bool(getattr(args, "no_proj", True)) if self.force_no_proj is None else self.force_no_proj
        self.part_pooling = getattr(args, "part_pooling", "first") if self.three_cat else "cls"
        self.encode_table = bool(getattr(args, "encode_table", True))
        if self.part_pooling[:3] == "att" and not (self.att and self.part_pooling == "att_self_pool"):
            # tb_retriever.py:433-434; 'att_self_multihead' needs attention.SelfAttention, which the reference does not ship
            raise NotImplementedError("part pooling only support att_self_pool (…ThreeCatAtt classes)")
        if self.att and self.part_pooling != "att_self_pool":
            raise NotImplementedError("part pooling only support att_self_pool, att_self_multihead is not built")
        if dtype is None:
            dtype = "bf16" if getattr(args, "fp16", False) else "fp32"
        if dtype not in ("fp32", "bf16"):
            raise ValueError("dtype must be 'fp32' or 'bf16'")
        self.dtype = dtype
        self.hidden_size = int(_cfg_get(config, "hidden_size"))
        self.device_index = None if device is None else torch.device(device).index
        pos_mode = _cfg_get(config, "position_mode")
        if pos_mode is None:
            pos_mode = "roberta" if self.roberta else "bert"
        pad = _cfg_get(config, "pad_token_id")
        if pad is None:
            pad = 1 if self.roberta else 0
        self._cfg = _lib.EncoderCfg(
            vocab_size=int(_cfg_get(config, "vocab_size")), hidden_size=self.hidden_size,
            num_hidden_layers=int(_cfg_get(config, "num_hidden_layers")),
            num_attention_heads=int(_cfg_get(config, "num_attention_heads")),
            intermediate_size=int(_cfg_get(config, "intermediate_size")),
            max_position_embeddings=int(_cfg_get(config, "max_position_embeddings")),
            type_vocab_size=int(_cfg_get(config, "type_vocab_size")),
            pad_token_id=int(pad),
            position_mode=_lib.POS_ROBERTA if pos_mode == "roberta" else _lib.POS_BERT,
            three_cat=1 if self.three_cat else 0, no_proj=1 if self.no_proj else 0,
            part_pooling=_lib.POOL.get(self.part_pooling, _lib.POOL["cls"]),
            layer_norm_eps=float(_cfg_get(config, "layer_norm_eps")))
        self._weights = {"encoder": OrderedDict(), "encoder_q": OrderedDict(), "project": OrderedDict()}
        self._handles = {}
        self.training = False

    # ---- nn.Module-like surface used by scripts/encode_corpus.py:98-99,130
    def _expected_keys(self):
        c = self._cfg
        keys = ["embeddings.word_embeddings.weight", "embeddings.position_embeddings.weight",
                "embeddings.token_type_embeddings.weight", "embeddings.LayerNorm.weight", "embeddings.LayerNorm.bias"]
        for i in range(c.num_hidden_layers):
            p = "encoder.layer.%d." % i
            for n in ("attention.self.query", "attention.self.key", "attention.self.value", "attention.output.dense",
                      "attention.output.LayerNorm", "intermediate.dense", "output.dense", "output.LayerNorm"):
                keys += [p + n + ".weight", p + n + ".bias"]
        keys += ["pooler.dense.weight", "pooler.dense.bias"]
        return keys

    def state_dict_keys(self):
        keys = [self.enc_prefix + "." + k for k in self._expected_keys()]
        if not self.shared_encoder:
            keys += ["encoder_q." + k for k in self._expected_keys()]
        keys += ["project.0.weight", "project.0.bias", "project.1.weight", "project.1.bias"]
        if self.att:
            keys += ["self_proj_tab.weight", "self_proj_tab.bias", "self_proj_psg.weight", "self_proj_psg.bias"]
        return set(keys)

    def state_dict(self):
        out = OrderedDict()
        for grp, prefix in (("encoder", self.enc_prefix), ("encoder_q", "encoder_q")):
            for k, v in self._weights[grp].items():
                out[prefix + "." + k] = v
        for k, v in self._weights["project"].items():
            out[k] = v
        return out

    def load_state_dict(self, state_dict, strict=True):
        have = self.state_dict_keys()
        unexpected = []
        for k, v in state_dict.items():
            if k.endswith(_IGNORED_SUFFIXES):
                continue
            if k not in have:
                unexpected.append(k)
                continue
            t = v.detach().to(torch.float32).cpu().contiguous()
            if k.startswith("encoder_q."):
                self._weights["encoder_q"][k[len("encoder_q."):]] = t
            elif k.startswith(self.enc_prefix + "."):
                self._weights["encoder"][k[len(self.enc_prefix) + 1:]] = t
            else:
                self._weights["project"][k] = t
        loaded = set(self.state_dict().keys())
        missing = sorted(k for k in have if k not in loaded and "pooler" not in k)
        if strict and (missing or unexpected):
            raise RuntimeError("load_state_dict: missing %r unexpected %r" % (missing[:5], unexpected[:5]))
        self._release()   # weights changed: rebuild device copies on next use
        return missing, unexpected

    def to(self, device):
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError("otter_amd encoders run on the GPU only (no CPU fallback)")
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        if idx != self.device_index:
            self._release()
            self.device_index = idx
        return self

    def cuda(self, device=None):
        return self.to("cuda" if device is None else device)

    def eval(self):
        self.training = False
        return self

    def half(self):
        self.dtype = "bf16"
        self._release()
        return self

    # ---- device handle
    def _release(self):
        lib = _lib.load() if self._handles else None
        for h in self._handles.values():
            lib.otter_encoder_destroy(h)
        self._handles = {}

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def _handle(self, which):
        if which in self._handles:
            return self._handles[which]
        lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.OtterError("no GPU visible: the encoder has no CPU fallback")
        if self.device_index is None:
            self.device_index = torch.cuda.current_device()
        h = ctypes.c_void_p()
        _lib.check(lib.otter_encoder_create(ctypes.byref(self._cfg), self.device_index, ctypes.byref(h)),
                   "otter_encoder_create")
        try:
            weights = dict(self._weights[which])
            if not weights:
                raise _lib.OtterError("no weights loaded for '%s' (use load_state_dict / load_saved)" % which)
            weights.update(self._weights["project"])
            for name, t in weights.items():
                arr = t.numpy()
                shape = (ctypes.c_int64 * arr.ndim)(*arr.shape)
                _lib.check(lib.otter_encoder_set_weight(h, name.encode(), ctypes.c_void_p(arr.ctypes.data), shape,
                                                        arr.ndim), "otter_encoder_set_weight(%s)" % name)
            _lib.check(lib.otter_encoder_finalize(h, _lib.OTTER_BF16 if self.dtype == "bf16" else _lib.OTTER_F32),
                       "otter_encoder_finalize")
        except Exception:
            lib.otter_encoder_destroy(h)
            raise
        self._handles[which] = h
        return h

    @staticmethod
    def _i64(t, dev, name):
        if t is None:
            return None
        if not torch.is_tensor(t):
            t = torch.as_tensor(np.asarray(t))
        if t.dim() != 2:
            raise ValueError("%s must be [B, L], got %r" % (name, tuple(t.shape)))
        return t.to(device=dev, dtype=torch.int64).contiguous()

    def _run(self, which, input_ids, attention_mask, token_type_ids, part2_mask, part3_mask):
        h = self._handle(which)
        dev = torch.device("cuda", self.device_index)
        ids = self._i64(input_ids, dev, "input_ids")
        B, L = ids.shape
        tensors = [ids]
        for name, t in (("input_mask", attention_mask), ("input_type_ids", token_type_ids),
                        ("part2_mask", part2_mask), ("part3_mask", part3_mask)):
            t = self._i64(t, dev, name)
            if t is not None and tuple(t.shape) != (B, L):
                raise ValueError("%s has shape %r, input_ids %r" % (name, tuple(t.shape), (B, L)))
            tensors.append(t)
        if tensors[1] is None:
            raise ValueError("input_mask is required")
        D = 3 * self.hidden_size if self.three_cat else self.hidden_size
        out = torch.empty((B, D), dtype=torch.float32, device=dev)
        ptr = [ctypes.c_void_p(t.data_ptr()) if t is not None else None for t in tensors]
        stream = ctypes.c_void_p(torch.cuda.current_stream(self.device_index).cuda_stream)
        _lib.check(_lib.load().otter_encoder_forward(h, ptr[0], ptr[1], ptr[2], ptr[3], ptr[4], B, L,
                                                     ctypes.c_void_p(out.data_ptr()), stream), "otter_encoder_forward")
        return out

    def hidden_states(self, input_ids, attention_mask, token_type_ids=None, n_layers=None):
        """Parity hook: last_hidden_state after n_layers layers, float32 [B, L, H]."""
        h = self._handle("encoder")
        dev = torch.device("cuda", self.device_index)
        ids = self._i64(input_ids, dev, "input_ids")
        mask = self._i64(attention_mask, dev, "input_mask")
        types = self._i64(token_type_ids, dev, "input_type_ids")
        B, L = ids.shape
        n = self._cfg.num_hidden_layers if n_layers is None else int(n_layers)
        out = torch.empty((B, L, self.hidden_size), dtype=torch.float32, device=dev)
        stream = ctypes.c_void_p(torch.cuda.current_stream(self.device_index).cuda_stream)
        _lib.check(_lib.load().otter_encoder_forward_hidden(
            h, ctypes.c_void_p(ids.data_ptr()), ctypes.c_void_p(mask.data_ptr()),
            ctypes.c_void_p(types.data_ptr()) if types is not None else None, B, L, n,
            ctypes.c_void_p(out.data_ptr()), stream), "otter_encoder_forward_hidden")
        return out

    # ---- the reference's methods
    def encode_seq(self, input_ids=None, attention_mask=None, token_type_ids=None, position_ids=None, head_mask=None,
                   output_hidden_states=True, part2_mask=None, part3_mask=None):
        if position_ids is not None or head_mask is not None:
            raise NotImplementedError("position_ids / head_mask are never passed by the reference's encode path")
        return self._run("encoder", input_ids, attention_mask, token_type_ids, part2_mask, part3_mask)

    def encode_q(self, input_ids=None, attention_mask=None, token_type_ids=None, position_ids=None, head_mask=None,
                 output_hidden_states=True, part2_mask=None, part3_mask=None):
        if position_ids is not None or head_mask is not None:
            raise NotImplementedError("position_ids / head_mask are never passed by the reference's encode path")
        which = "encoder" if self.shared_encoder else "encoder_q"
        return self._run(which, input_ids, attention_mask, token_type_ids, part2_mask, part3_mask)

    def forward(self, batch):
        fn = self.encode_seq if self.encode_table else self.encode_q
        types = None if self.roberta else batch.get("input_type_ids")
        if self.three_cat:
            cls = fn(batch["input_ids"], batch["input_mask"], types, part2_mask=batch["part2_mask"],
                     part3_mask=batch["part3_mask"])
        else:
            cls = fn(batch["input_ids"], batch["input_mask"], types)
        return {"embed": cls}

    __call__ = forward


class DataParallel:
    """model = torch.nn.DataParallel(model) of scripts/encode_corpus.py:112-113: ONE process, a batch cut along its
    first dimension over the GPUs, results gathered on device_ids[0].  nn.DataParallel broadcasts the weights and
    rebuilds its replicas on every call; here each GPU keeps its own resident copy (built on first use from the
    module's host weights) and a call only moves the token tensors out and the embeddings back.  Every replica's
    forward is only enqueued on its GPU's current stream, so the GPUs run side by side under one host thread.

    device_ids: HIP ordinals (default: every visible GPU); an ordinal may repeat (the one-GPU tests).  The object
    has the module's surface (`eval()`, `forward(batch)`, `module`, `device_index` = device_ids[0]), so
    encode_driver.predict pipelines it like a single encoder.  One process per GPU instead: sharded.encode_rows."""

    def __init__(self, module, device_ids=None):
        if device_ids is None:
            device_ids = list(range(torch.cuda.device_count()))
        device_ids = [int(d) for d in device_ids]
        if not device_ids:
            raise ValueError("DataParallel: no devices")
        self.module = module
        self.device_ids = device_ids
        self.device_index = device_ids[0]
        module.to("cuda:%d" % device_ids[0])
        self.replicas = [module]
        for d in device_ids[1:]:
            r = type(module)(module.config, module.args, device="cuda:%d" % d, dtype=module.dtype)
            r._weights = module._weights            # the same host tensors: no copy until the replica's first call
            self.replicas.append(r)
        self.training = False
        # debug counter: host time spent enqueueing each replica's share of a call (scatter copy + forward launches),
        # seconds, and the calls counted.  The GPUs run side by side only while G x (enqueue time per replica) stays
        # below the device time of one share: enqueue_report() says how far that is.
        self.enqueue_s = [0.0] * len(self.replicas)
        self.enqueue_calls = 0

    def enqueue_report(self):
        """{'per_replica_ms': [...], 'calls': n, 'sum_ms': host ms per call over all replicas}: the host-side cost of one
        forward() call; compare sum_ms with the device time of one replica's share (14 ms at 128 x 512)."""
        n = max(1, self.enqueue_calls)
        per = [1e3 * t / n for t in self.enqueue_s]
        return {"per_replica_ms": per, "calls": self.enqueue_calls, "sum_ms": sum(per)}

    def eval(self):
        for r in self.replicas:
            r.eval()
        return self

    def to(self, device):
        if torch.device(device).index not in (None, self.device_ids[0]):
            raise RuntimeError("DataParallel: the replicas stay on device_ids; results are gathered on cuda:%d" % self.device_ids[0])
        return self

    def state_dict(self):
        return OrderedDict(("module." + k, v) for k, v in self.module.state_dict().items())

    def load_state_dict(self, state_dict, strict=True):
        """Keys with or without nn.DataParallel's `module.` prefix; every replica is rebuilt on its next call."""
        sd = OrderedDict((k[7:] if k.startswith("module.") else k, v) for k, v in state_dict.items())
        res = self.module.load_state_dict(sd, strict=strict)
        for r in self.replicas[1:]:
            r._weights = self.module._weights
            r._release()
        return res

    def forward(self, batch):
        first = next(t for t in batch.values() if torch.is_tensor(t))
        B = int(first.shape[0])
        G = len(self.replicas)
        per = -(-B // G)                                 # nn.DataParallel's scatter: chunks of ceil(B / G)
        outs = []
        for g, r in enumerate(self.replicas):
            lo, hi = g * per, min(B, (g + 1) * per)
            if hi <= lo:
                break
            dev = torch.device("cuda", self.device_ids[g])
            t0 = time.perf_counter()
            sub = {k: (v[lo:hi].to(dev, non_blocking=True) if torch.is_tensor(v) else v) for k, v in batch.items()}
            with torch.cuda.device(dev):
                outs.append(r(sub)["embed"])
            self.enqueue_s[g] += time.perf_counter() - t0
        self.enqueue_calls += 1
        home = torch.device("cuda", self.device_ids[0])
        if len(outs) == 1:
            return {"embed": outs[0].to(home)}
        return {"embed": torch.cat([o.to(home, non_blocking=True) for o in outs], dim=0)}

    __call__ = forward


class SingleEncoderThreeCatPool(_EncoderBase):
    """retrieval/models/tb_retriever.py:150-227 (BERT family: token types, arange positions)."""
    three_cat, roberta = True, False


class RobertaSingleEncoderThreeCatPool(_EncoderBase):
    """retrieval/models/tb_retriever.py:250-261 — the reference's default encoder."""
    three_cat, roberta = True, True


class SingleEncoderThreeCatAtt(_EncoderBase):
    """retrieval/models/tb_retriever.py:404-523 with part_pooling='att_self_pool' (BERT family)."""
    three_cat, roberta, att = True, False, True


class RobertaSingleEncoderThreeCatAtt(_EncoderBase):
    """retrieval/models/tb_retriever.py:555-560 — chosen by scripts/encode_corpus.py:50-52 for 'att*' pooling."""
    three_cat, roberta, att = True, True, True


class SingleEncoder(_EncoderBase):
    """retrieval/models/retriever.py:249-311."""
    three_cat, roberta = False, False


class RobertaSingleEncoder(_EncoderBase):
    """retrieval/models/retriever.py:334-345."""
    three_cat, roberta = False, True


class RobertaCtxEncoder(_EncoderBase):
    """retrieval/models/retriever.py:547-564: one vector, the projection head always on
    (`project(encoder(ids, mask)[0][:, 0, :])`), whatever args.no_proj says."""
    three_cat, roberta, force_no_proj = False, True, False

    def forward(self, batch):
        return {"embed": self.encode_seq(batch["input_ids"], batch["input_mask"])}

    __call__ = forward


class CtxEncoder(_EncoderBase):
    """retrieval/models/retriever.py:513-544 (BERT family): the HF model is the attribute `encoder_c`
    (state_dict keys 'encoder_c.*'), no projection, embed = h[:, 0].  Only args.multi_vector <= 1 is built: the
    reference's multi-vector branches read `self.encoder`, an attribute that class never sets."""
    three_cat, roberta, force_no_proj, enc_prefix = False, False, True, "encoder_c"

    def __init__(self, config, args, device=None, dtype=None):
        if int(getattr(args, "multi_vector", 1) or 1) > 1:
            raise NotImplementedError("CtxEncoder: multi_vector > 1 reads self.encoder, which the reference class does not define")
        super().__init__(config, args, device=device, dtype=dtype)

    def forward(self, batch):
        return {"embed": self.encode_seq(batch["input_ids"], batch["input_mask"], batch.get("input_type_ids"))}

    __call__ = forward
