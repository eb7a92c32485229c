"""Fused optimizer step for the decoder: the reference's vendored RAdam
(radam.py:63-142, selected by `optim_algo: RAdam`, tts_lightning_modules.py:557-559) and
Lightning's `gradient_clip_val: 1.0 / gradient_clip_algorithm: norm`
(configs/RADMMM_train_config.yaml:7-8) on flat fp32 buffers.

Parameters are re-pointed to views of one flat buffer per bucket (the same buckets
`ddp.BucketedGradReducer` uses for the gradients), so one step is one streaming HIP kernel per
bucket instead of ~10 torch kernels for each of the ~300 parameter tensors; the clip
coefficient stays on the device (no host sync).  state_dict()/load_state_dict() use the
reference's layout (`state[p] = {step, exp_avg, exp_avg_sq}`)."""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Callable, Dict, Iterable, List, Optional, Tuple

import torch

from ._lib import lib, check, ptr, stream
from .ddp import default_bucket_key, slot_numel


class FlatRAdam:
    """RAdam(params, lr, betas, eps, weight_decay) of the reference, flat and fused.

    named_params: iterable of (name, Parameter) (e.g. module.named_parameters()); parameters with
    requires_grad False are ignored.  If `reducer` (a BucketedGradReducer over the same module) is
    given its flat gradient buckets are used, otherwise gradients are gathered into own flats.

    Deliberate deviations from radam.py (both invisible when every parameter gets a gradient every step, which holds for
    the decoder): a parameter whose .grad is None is treated as having a ZERO gradient (its moments decay and weight decay
    applies; the reference skips it entirely, radam.py `if p.grad is None: continue`), and there is one step counter for
    all parameters (the reference counts per parameter; load_state_dict takes the last entry's `step`)."""

    def __init__(self, named_params: Iterable[Tuple[str, torch.nn.Parameter]], lr: float = 1e-3,
                 betas: Tuple[float, float] = (0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
                 bucket_key: Callable[[str], str] = default_bucket_key, reducer=None):
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.step_count = 0
        groups: "OrderedDict[str, List[torch.nn.Parameter]]" = OrderedDict()
        self._order: List[torch.nn.Parameter] = []       # the caller's parameter order = the reference optimizer's indices
        for name, p in named_params:
            if p.requires_grad:
                assert p.dtype == torch.float32 and p.is_cuda, "FlatRAdam: fp32 parameters on the GPU"
                groups.setdefault(bucket_key(name), []).append(p)
                self._order.append(p)
        self.buckets: List[Dict] = []
        red_flats = {b["key"]: b for b in reducer.buckets} if reducer is not None else {}
        for key, params in groups.items():
            gflat = None
            if key in red_flats:
                rb = red_flats[key]
                assert sorted(id(q) for q in rb["params"]) == sorted(id(q) for q in params), "bucket contents differ"
                params = list(rb["params"])             # the reducer's order (direct-write parameters first)
                gflat = rb["flat"]
            n = sum(slot_numel(p) for p in params)           # 16-byte aligned slots: the reducer's layout (ddp.slot_numel)
            flat = torch.zeros(n, device=params[0].device, dtype=torch.float32)
            off = 0
            for p in params:
                flat[off: off + p.numel()].copy_(p.data.reshape(-1))
                p.data = flat[off: off + p.numel()].view_as(p)          # parameter becomes a view of the flat buffer
                off += slot_numel(p)
            self.buckets.append(dict(key=key, params=params, flat=flat, gflat=gflat, own_g=gflat is None,
                                     m=torch.zeros_like(flat), v=torch.zeros_like(flat)))
        self._slot = {}                                    # id(param) -> (bucket, offset)
        for b in self.buckets:
            off = 0
            for p in b["params"]:
                self._slot[id(p)] = (b, off)
                off += slot_numel(p)
        dev = self.buckets[0]["flat"].device
        self._part = torch.empty(len(self.buckets), int(lib.radmmm_sumsq_scratch_floats()), device=dev)
        self._clip = torch.ones(1, device=dev)

    # -- gradients --------------------------------------------------------------------------------
    def _gather_grads(self):
        for b in self.buckets:
            if not b["own_g"]:
                continue
            if b["gflat"] is None:
                b["gflat"] = torch.zeros_like(b["flat"])
            off = 0
            for p in b["params"]:
                if p.grad is not None:
                    b["gflat"][off: off + p.numel()].copy_(p.grad.reshape(-1))
                else:
                    b["gflat"][off: off + p.numel()].zero_()
                off += slot_numel(p)

    def clip_grad_norm(self, max_norm: float) -> torch.Tensor:
        """Global 2-norm of all gradients (device scalar, returned) and the clip coefficient
        min(1, max_norm / (norm + 1e-6)) kept on the device for the next step()."""
        self._gather_grads()
        for i, b in enumerate(self.buckets):
            check(lib.radmmm_sumsq(ptr(b["gflat"]), b["gflat"].numel(), ptr(self._part[i]), stream()), "sumsq")
        total = self._part.double().sum().sqrt().float()
        self._clip = torch.clamp(max_norm / (total + 1e-6), max=1.0).reshape(1).contiguous()
        self._clipped = True
        return total

    # -- update -----------------------------------------------------------------------------------
    def step(self):
        if not getattr(self, "_clipped", False):
            self._gather_grads()
            self._clip = None
        self._clipped = False
        self.step_count += 1
        t = self.step_count
        beta1, beta2 = self.betas
        beta2_t = beta2 ** t
        n_sma_max = 2.0 / (1.0 - beta2) - 1.0
        n_sma = n_sma_max - 2.0 * t * beta2_t / (1.0 - beta2_t)
        if n_sma >= 5:
            step_size = self.lr * math.sqrt((1 - beta2_t) * (n_sma - 4) / (n_sma_max - 4) * (n_sma - 2) / n_sma
                                            * n_sma_max / (n_sma_max - 2)) / (1 - beta1 ** t)
        else:
            step_size = self.lr / (1 - beta1 ** t)
        for b in self.buckets:
            check(lib.radmmm_radam_step(ptr(b["flat"]), ptr(b["gflat"]), ptr(b["m"]), ptr(b["v"]), b["flat"].numel(),
                                        ptr(self._clip), beta1, beta2, self.eps, step_size, self.weight_decay * self.lr,
                                        1 if n_sma >= 5 else 0, stream()), "radam_step")

    def zero_grad(self, set_to_none: bool = True):
        """torch.optim.Optimizer.zero_grad semantics for the standard loop `opt.zero_grad(); loss.backward(); opt.step()`:
        without a reducer the parameters' own .grad tensors are what autograd accumulates into, so they are reset here
        (the flat copies are refilled from them by step()); with a reducer the buckets ARE the gradients and
        reducer.prepare() re-arms them."""
        for b in self.buckets:
            if b["gflat"] is not None and b["own_g"]:
                b["gflat"].zero_()
            if b["own_g"]:
                for p in b["params"]:
                    if p.grad is not None:
                        if set_to_none:
                            p.grad = None
                        else:
                            p.grad.zero_()

    # -- reference-compatible state ---------------------------------------------------------------
    def state_dict(self):
        state = {}
        for idx, p in enumerate(self._order):
            b, off = self._slot[id(p)]
            n = p.numel()
            state[idx] = {"step": self.step_count, "exp_avg": b["m"][off: off + n].view_as(p).clone(),
                          "exp_avg_sq": b["v"][off: off + n].view_as(p).clone()}
        return {"state": state, "param_groups": [{"lr": self.lr, "betas": self.betas, "eps": self.eps,
                                                  "weight_decay": self.weight_decay, "params": list(range(len(self._order)))}]}

    def load_state_dict(self, sd):
        for idx, p in enumerate(self._order):
            st = sd["state"].get(idx)
            if st is None:
                continue
            b, off = self._slot[id(p)]
            n = p.numel()
            b["m"][off: off + n].copy_(st["exp_avg"].reshape(-1))
            b["v"][off: off + n].copy_(st["exp_avg_sq"].reshape(-1))
            self.step_count = int(st["step"])
        g = sd["param_groups"][0]
        self.lr, self.betas, self.eps, self.weight_decay = g["lr"], tuple(g["betas"]), g["eps"], g["weight_decay"]
