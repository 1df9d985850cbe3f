"""Drop-in for the reference's ``models.line_transformer`` module.

``LineTransformer`` keeps the reference's constructor/config/``preprocess``/``forward``/
``subline2keyline``/``default_ret`` surface and its ``state_dict`` key layout (so the authors'
``LineTR_weight.pth`` loads strictly), but owns no arithmetic: its nn.Module tree only HOLDS the
parameters; tokenisation, the descriptor network and the matcher run in liblinetr_hip.so through
``linetr_amd.engine.Engine``.  There is no CPU execution path -- calling it without a HIP device raises.

Reference behaviour mirrored here (file:line in the reference checkout):
  * config dict merged over default_config and mutated by callers (models/line_transformer.py:187-206)
  * mode == 'test' loads weights/LineTR_weight.pth (next to the `models/` shim, or in this package) strictly and prints a message (:220-223)
  * preprocess(): cv2 KeyLines -> arrays -> remove_borders -> filter_by_length -> tokeniser (:251-275; the module-level
    functions live in linetr_amd.line_process, as in the reference);
    writes config['image_shape'] = image_shape (:258); ndarray valid masks honoured, tensors ignored
  * forward(): same dict object returned with 'line_desc' [1,256,N] added (:225-249)
"""
from __future__ import annotations

import operator
from pathlib import Path

import numpy as np
import torch
from torch import nn

from .engine import Engine
from .line_process import *  # noqa: F401,F403  (the reference's module does the same: models/line_transformer.py:6)
from .line_process import (_token_engine, change_cv2_T_np, filter_by_length, get_angles, get_dist_matrix,  # noqa: F401
                           prefilter_tokenize, remove_borders, tokenize_into)

_VERSION_OF = operator.attrgetter("_version")

__all__ = ["LineTransformer", "get_dist_matrix", "change_cv2_T_np", "remove_borders", "filter_by_length",
           "get_angles", "get_line_dist", "point_on_line", "sample_descriptors", "line_tokenizer", "preprocess"]


# ------------------------------------------------------------------------------------------------
# parameter containers with the reference's state_dict keys
# ------------------------------------------------------------------------------------------------

def _pointwise_stack(widths):
    mods = []
    for i, (a, b) in enumerate(zip(widths[:-1], widths[1:])):
        mods.append(nn.Conv1d(a, b, kernel_size=1))
        if i < len(widths) - 2:
            mods += [nn.BatchNorm1d(b), nn.ReLU()]
    nn.init.zeros_(mods[-1].bias)
    return nn.Sequential(*mods)


class _Holder(nn.Module):
    """A module that only groups parameters (no forward of its own)."""

    def __init__(self, **children):
        super().__init__()
        for k, v in children.items():
            setattr(self, k, v)


def _param_tree(d, enc, heads, n_desc, d_inner, n_sig):
    def desc_layer():
        att = _Holder(w_qs=nn.Linear(d, d), w_ks=nn.Linear(d, d), w_vs=nn.Linear(d, d), fc=nn.Linear(d, d),
                      layer_norm=nn.LayerNorm(d, eps=1e-6))
        ffn = _Holder(w_1=nn.Linear(d, d_inner), w_2=nn.Linear(d_inner, d), layer_norm=nn.LayerNorm(d, eps=1e-6))
        return _Holder(slf_attn=att, pos_ffn=ffn)

    def sig_layer():
        merge = nn.Conv1d(d, d, kernel_size=1)
        proj = nn.ModuleList([nn.Conv1d(d, d, kernel_size=1) for _ in range(3)])
        for p in proj:
            p.load_state_dict(merge.state_dict())
        return _Holder(attn=_Holder(merge=merge, proj=proj), mlp=_pointwise_stack([2 * d, 2 * d, d]))

    klenc = _Holder(line_position_enc=_Holder(encoder=_pointwise_stack([5, *enc, d])),
                    word_position_enc=_Holder(encoder=_pointwise_stack([3, *enc, d])),
                    desc_layers=nn.ModuleList([desc_layer() for _ in range(n_desc)]))
    klenc.cls_token = nn.Parameter(torch.randn(1, 1, 1, d))
    selfattn = _Holder(layers=nn.ModuleList([sig_layer() for _ in range(n_sig)]))
    return klenc, selfattn, nn.Conv1d(d, d, kernel_size=1)


class LineTransformer(nn.Module):
    """Line-Transformer descriptor network behind the reference's call surface."""

    default_config = {
        "mode": "test",
        "image_shape": [480, 640],
        "min_length": 16,
        "token_distance": 8,
        "max_tokens": 21,
        "remove_borders": 8,
        "max_keylines": -1,
        "descriptor_dim": 256,
        "keyline_encoder": [32, 64, 128, 256],
        "n_heads": 4,
        "n_line_descriptive_layers": 1,
        "d_inner": 1024,
    }
    N_SIGNATURE_LAYERS = 7
    # The reference hard-codes dropout 0.1 in its attention blocks (models/line_transformer.py:77,96; models/line_attention.py:8,25,79):
    # active whenever the module is in train mode.  The training-time forward of this build is deterministic -- BatchNorm on batch
    # statistics, dropout probability 0, no autograd -- so it only runs once the caller has said so by setting
    # `model.dropout = 0.0`, the counterpart of zeroing `p` on the reference's nn.Dropout instances.
    dropout = 0.1

    def __init__(self, config):
        super().__init__()
        self.config = {**self.default_config, **config}
        self.image_shape = self.config["image_shape"]
        c = self.config
        self.klenc, self.selfattn, self.final_proj = _param_tree(
            c["descriptor_dim"], c["keyline_encoder"], c["n_heads"], c["n_line_descriptive_layers"], c["d_inner"],
            self.N_SIGNATURE_LAYERS)
        self._engine = None
        self._engine_key = None
        if c["mode"] == "test":
            self.load_state_dict(torch.load(self._weight_file()))
            print("Loaded Line-Transformer model")

    @staticmethod
    def _weight_file():
        """The authors' checkpoint.  The reference loads <its models package>/weights/LineTR_weight.pth
        (models/line_transformer.py:220-221); after the drop-in that file still sits next to the `models/` shim, so it is looked
        for there first, then in this package, then under $LINETR_WEIGHTS."""
        import os
        import sys
        cands = []
        shim = sys.modules.get("models")
        for base in list(getattr(shim, "__path__", [])) + [str(Path(__file__).parent)]:
            cands.append(Path(base) / "weights/LineTR_weight.pth")
        if os.environ.get("LINETR_WEIGHTS"):
            cands.append(Path(os.environ["LINETR_WEIGHTS"]))
        for p in cands:
            if p.exists():
                return p
        raise FileNotFoundError("LineTR_weight.pth not found; looked in: " + ", ".join(str(p) for p in cands))

    # -- native engine management ---------------------------------------------------------------
    def _device(self):
        return self.final_proj.weight.device

    def load_state_dict(self, *a, **k):
        self._engine = None
        self.__dict__["_engine_train"] = None
        self.__dict__["_tracked"] = None
        return super().load_state_dict(*a, **k)

    def _apply(self, fn, *a, **k):
        self._engine = None
        self.__dict__["_engine_train"] = None
        self.__dict__["_tracked"] = None
        return super()._apply(fn, *a, **k)

    def _params_version(self):
        """_weights_version without the buffers: the training-mode engine keeps the convolutions unfolded and never reads the
        BatchNorm running statistics, which every training-time forward rewrites."""
        ps = [p for p in self.parameters()]
        return tuple(id(p) for p in ps) + tuple(map(torch.Tensor.data_ptr, ps)) + tuple(p._version for p in ps if not p.is_inference())

    def _weights_version(self):
        """Changes whenever a parameter / buffer is modified in place through the tensor itself (optimizer step, `with
        torch.no_grad(): p.copy_(..)`, fine-tuning; writes through the detached alias `p.data` carry their own counter and are
        NOT seen: call load_state_dict or `.to()` after those), re-bound
        (p.data = t) or replaced (module.weight = nn.Parameter(...)): the native engine holds a folded COPY of the weights
        and must be rebuilt then.  The module tree is walked once per engine; every call afterwards re-checks the cached
        (owner dict, name, tensor) triples -- identity, storage pointer, version counter -- which costs ~40 us instead of the
        ~340 us of a fresh named_parameters() walk (it runs on every forward of the drop-in path)."""
        track = self.__dict__.get("_tracked")
        if track is None:
            owners, names, tensors = [], [], []
            n_entries, dicts = 0, []
            for mod in self.modules():
                dicts += [mod._parameters, mod._buffers, mod._modules]
                for d in (mod._parameters, mod._buffers):
                    n_entries += len(d)
                    for name, t in d.items():
                        if t is not None:
                            owners.append(d); names.append(name); tensors.append(t)
                # the sub-module objects themselves: `lt.final_proj = nn.Conv1d(..)` or a swapped encoder block leaves the OLD
                # module's parameter dicts untouched, so the identity sweep has to see the parent's _modules entry change
                n_entries += len(mod._modules)
                for name, child in mod._modules.items():
                    if child is not None:
                        owners.append(mod._modules); names.append(name); tensors.append(child)
            params = [t for t in tensors if isinstance(t, torch.Tensor)]
            versioned = [t for t in params if not t.is_inference()]          # inference tensors keep no version counter
            track = self.__dict__["_tracked"] = (owners, names, tensors, versioned, params, dicts, n_entries)
        owners, names, tensors, versioned, params, dicts, n_entries = track
        # C-level sweeps over the ~300 tracked objects (a Python loop with the same reads costs twice as much); the entry count
        # catches a parameter / buffer / sub-module ADDED to or deleted from a tracked dict
        if sum(map(len, dicts)) != n_entries or not all(map(operator.is_, map(dict.get, owners, names), tensors)):
            self.__dict__["_tracked"] = None      # a replaced object: walk the tree again (its storage pointer is in the new key)
            return self._weights_version()
        # data_ptr: `p.data = new_tensor` rebinds storage without touching _version
        return tuple(map(torch.Tensor.data_ptr, params)) + tuple(map(_VERSION_OF, versioned))

    # the native handle is a ctypes pointer: never pickled / deep-copied, rebuilt on first use instead
    def __getstate__(self):
        state = self.__dict__.copy()
        state["_engine"] = None
        state["_engine_key"] = None
        state["_engine_train"] = None
        state["_engine_train_key"] = None
        state["_tracked"] = None
        return state

    def __deepcopy__(self, memo):
        import copy
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            new.__dict__[k] = None if k in ("_engine", "_engine_key", "_engine_train", "_engine_train_key", "_tracked") else copy.deepcopy(v, memo)
        return new

    def _bn_layers(self):
        """The 15 BatchNorm1d modules in the order of the native packed statistics (word encoder, line encoder, signature MLPs)."""
        mods = [m for enc in (self.klenc.word_position_enc.encoder, self.klenc.line_position_enc.encoder)
                for m in enc if isinstance(m, nn.BatchNorm1d)]
        return mods + [layer.mlp[1] for layer in self.selfattn.layers]

    def _forward_train(self, data):
        """train.py:127,163-164: the module in train mode, called on a batch of fixed-size samples.  Every BatchNorm1d normalises with
        the statistics of THIS call's batch and moves its running_mean / running_var / num_batches_tracked exactly as
        torch.nn.BatchNorm1d does (linetr_forward_train, csrc/lt_bntrain.h).  Forward only: the returned line_desc carries no
        autograd graph, and dropout must have been switched off (`model.dropout = 0.0`)."""
        if self.dropout != 0:
            raise RuntimeError(
                "linetr_amd.LineTransformer in train mode: the training-time forward of this build has no dropout (the reference "
                "applies nn.Dropout(0.1) in its attention blocks, models/line_attention.py:11,39,84) and no autograd.  Set "
                "`model.dropout = 0.0` to run the deterministic train-mode forward (BatchNorm batch statistics, running statistics "
                "updated), or call .eval() for inference (match_line_pairs.py:75, demo_LineTR.py:154, train.py:133)")
        bns = self._bn_layers()
        for bn in bns:
            if bn.momentum is None or bn.eps != 1e-5 or not bn.track_running_stats or not bn.affine:
                raise RuntimeError("linetr_amd train-mode forward: BatchNorm1d must keep the reference's settings "
                                   "(momentum given, eps 1e-5, affine, track_running_stats)")
        momentum = float(bns[0].momentum)
        if any(float(bn.momentum) != momentum for bn in bns):
            raise RuntimeError("linetr_amd train-mode forward: one momentum for all BatchNorm layers")
        sub = data["sublines"]
        dev = sub.device if sub.is_cuda else self._device()
        if dev.type != "cuda":
            raise RuntimeError("LineTransformer (linetr_amd) runs on a HIP device only: move the module with .to('cuda'); there is no CPU fallback")
        if dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        key = (dev, tuple(self.image_shape[-2:]), self._params_version())
        if self.__dict__.get("_engine_train") is None or self.__dict__.get("_engine_train_key") != key:
            c = self.config
            self.__dict__["_engine_train"] = Engine(self.state_dict(), dev, descriptor_dim=c["descriptor_dim"],
                                                    keyline_encoder=list(c["keyline_encoder"]), n_heads=c["n_heads"],
                                                    n_line_descriptive_layers=c["n_line_descriptive_layers"], d_inner=c["d_inner"],
                                                    n_sig_layers=self.N_SIGNATURE_LAYERS, image_shape=list(self.image_shape[-2:]),
                                                    bn_batch_stats=True)
            self.__dict__["_engine_train_key"] = key
        eng = self.__dict__["_engine_train"]
        B, N = int(sub.shape[0]), int(sub.shape[1])
        T = int(data["pnt_sublines"].shape[2])
        flat = lambda t, *tail: t.reshape(B * N, *tail)
        with torch.no_grad():
            running = torch.cat([t.detach().to(device=dev, dtype=torch.float32).reshape(-1) for bn in bns
                                 for t in (bn.running_mean, bn.running_var)])
            out = eng.forward_train_tensors(flat(sub, 2, 2), flat(data["pnt_sublines"], T, 2), flat(data["resp_sublines"]),
                                            flat(data["angle_sublines"], 2), flat(data["desc_sublines"], T, 256),
                                            flat(data["score_sublines"], T), np.arange(B + 1, dtype=np.int32) * N, running,
                                            momentum=momentum)
            off = 0
            for bn in bns:     # the updated statistics go back through copy_ (version counters move: an eval engine is rebuilt)
                C_ = bn.num_features
                bn.running_mean.copy_(running[off:off + C_]); bn.running_var.copy_(running[off + C_:off + 2 * C_])
                bn.num_batches_tracked.add_(1)
                off += 2 * C_
        data.update({"line_desc": out.view(B, N, 256).transpose(1, 2)})
        return data

    def engine(self, device=None) -> Engine:
        if self.training:
            # the inference engine folds BatchNorm(eval) into the convolutions: a module in train mode goes through _forward_train
            # (BatchNorm on batch statistics); anything that asks for THIS engine would silently get validation-mode descriptors
            raise RuntimeError("linetr_amd.LineTransformer is in train mode: the inference engine (BatchNorm running statistics) "
                               "is not served; forward() runs the training-time path, everything else needs .eval() "
                               "(match_line_pairs.py:75, demo_LineTR.py:154, train.py:133)")
        dev = torch.device(device) if device is not None else self._device()
        if dev.type != "cuda":
            raise RuntimeError("LineTransformer (linetr_amd) runs on a HIP device only: move the module with "
                               ".to('cuda'); there is no CPU fallback")
        if dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        key = (dev, tuple(self.image_shape[-2:]), self._weights_version())
        if self._engine is None or self._engine_key != key:
            c = self.config
            self._engine = Engine(self.state_dict(), dev, descriptor_dim=c["descriptor_dim"],
                                  keyline_encoder=list(c["keyline_encoder"]), n_heads=c["n_heads"],
                                  n_line_descriptive_layers=c["n_line_descriptive_layers"], d_inner=c["d_inner"],
                                  n_sig_layers=self.N_SIGNATURE_LAYERS, image_shape=list(self.image_shape[-2:]))
            self._engine_key = key
        return self._engine

    # -- reference surface ------------------------------------------------------------------------
    def preprocess(self, klines_cv, image_shape, pred_superpoint, valid_mask=None):
        """Line tokenisation.  Returns the reference's dict (11 tensor entries, leading batch axis 1)."""
        _, _, height, width = self.config["image_shape"] = image_shape
        # (the reference builds np.ones((height, width)) for a missing mask, :261-262; an all-ones mask keeps every line, so the
        # 2.4 MB array is simply not made.)  The tokeniser needs no weights: the weight-less engine of line_process serves it.
        c = self.config
        return prefilter_tokenize(klines_cv, height, width, c["remove_borders"], c["min_length"], c["max_keylines"],
                                  c["token_distance"], c["max_tokens"], pred_superpoint, valid_mask)

    def forward(self, data):
        if len(data["klines"]) == 0:
            return self.default_ret()
        if self.training:
            return self._forward_train(data)
        sub = data["sublines"]
        B, N = int(sub.shape[0]), int(sub.shape[1])
        T = int(data["pnt_sublines"].shape[2])
        eng = self.engine(sub.device if sub.is_cuda else None)
        flat = lambda t, *tail: t.reshape(B * N, *tail)
        out = eng.forward_tensors(flat(sub, 2, 2), flat(data["pnt_sublines"], T, 2), flat(data["resp_sublines"]),
                                  flat(data["angle_sublines"], 2), flat(data["desc_sublines"], T, 256),
                                  flat(data["score_sublines"], T), np.arange(B + 1, dtype=np.int32) * N)
        data.update({"line_desc": out.view(B, N, 256).transpose(1, 2)})
        return data

    def forward_many(self, datas):
        """forward() for several pre-processed images in ONE native call (a var-len batch: the descriptor network of an image never
        looks at another image, so every dict receives exactly the 'line_desc' forward() would give it, up to fp32 round-off --
        tests/test_gpu_properties.py).  Halves the launches of a pair; Matching.forward uses it for its two images.  Dicts
        without lines get default_ret(), like forward()."""
        live = [d for d in datas if len(d["klines"]) != 0]
        outs = {id(d): self.default_ret() for d in datas if len(d["klines"]) == 0}
        if self.training:                        # batch statistics are per call: one call per dict, as separate forward()s would be
            for d in live:
                outs[id(d)] = self.forward(d)
        elif len(live) == 1:
            outs[id(live[0])] = self.forward(live[0])
        elif live:
            T = int(live[0]["pnt_sublines"].shape[2])
            if any(int(d["sublines"].shape[0]) != 1 or int(d["pnt_sublines"].shape[2]) != T for d in live):
                for d in live:                       # batched / differently tokenised inputs: one call each
                    outs[id(d)] = self.forward(d)
            else:
                n = [int(d["sublines"].shape[1]) for d in live]
                cu = np.concatenate([[0], np.cumsum(n)]).astype(np.int32)
                cat = lambda k, *tail: torch.cat([d[k].reshape(m, *tail) for d, m in zip(live, n)])
                eng = self.engine(live[0]["sublines"].device if live[0]["sublines"].is_cuda else None)
                ld = eng.forward_tensors(cat("sublines", 2, 2), cat("pnt_sublines", T, 2), cat("resp_sublines"),
                                         cat("angle_sublines", 2), cat("desc_sublines", T, 256), cat("score_sublines", T), cu)
                for i, d in enumerate(live):
                    d.update({"line_desc": ld[cu[i]:cu[i + 1]].t()[None]})
                    outs[id(d)] = d
        return [outs[id(d)] for d in datas]

    def subline2keyline(self, distance_sublines, mat_klines2sublines0, mat_klines2sublines1):
        """Mean sub-line distance per key-line pair: (A0 @ D @ A1^T)[None], NumPy in / NumPy out
        (models/line_transformer.py:277-282).  The call sites hand over plain [K,N] matrices (matching.py:80 indexes [0]), so the
        matrices are read by their contents, on the device: a tokeniser's matrix (one non-zero per column, rows of
        1 / num_sublines) is reduced to its sub-line -> key-line map and pooled by the matcher's segmented-mean kernel, any other
        matrix is multiplied out as given (linetr_pool_distmat_dense).  No torch arithmetic."""
        a0, a1 = mat_klines2sublines0, mat_klines2sublines1
        dev = next((t.device for t in (a0, a1) if torch.is_tensor(t) and t.is_cuda), None)
        eng = _token_engine(dev if dev is not None else self._device())       # the pooling needs no weights
        d = torch.as_tensor(np.asarray(distance_sublines), dtype=torch.float32)
        if d.dim() != 2:
            raise ValueError("subline2keyline: distance_sublines must be [N0,N1]")
        a0, a1 = (torch.as_tensor(a, dtype=torch.float32) for a in (a0, a1))
        if a0.dim() != 2 or a1.dim() != 2:
            raise ValueError("subline2keyline: mat_klines2sublines must be [K,N] (the reference's callers index the batch axis away)")
        return eng.pool_distmat_dense(d, a0, a1)[None].cpu().numpy()

    def default_ret(self):
        return {"klines": torch.empty((1, 0, 2, 2)), "sublines": torch.empty((1, 0, 2, 2)),
                "line_desc": torch.empty((1, 256, 0)), "mat_klines2sublines": torch.empty((1, 0, 0))}
