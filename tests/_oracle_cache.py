"""Session cache of full-size CPU-oracle runs (forward + NLL + whole backward of the decoder).  Several GPU tests compare
different product schemes / kernels against the oracle on THE SAME seeded batch and weights; the oracle result depends on
neither, so it is computed once per (config, batch shape, seed) and shared.  Test infrastructure only."""
import numpy as np
import torch

_CACHE = {}


def _T(d):
    return {k: torch.from_numpy(np.asarray(v)) for k, v in d.items()}


def oracle_decoder_run(kw, B, Tn, seed, ragged=True):
    """{"sd", "batch", "z_mel", "log_det_W_list", "log_s_sums", "loss", "grads" (per parameter name), "g_mel", "g_ctx"} for
    the procedural weights of DecoderConfig(**kw) on synthetic_batch(B, Tn, seed, ragged).  Tensors are detached."""
    key = (tuple(sorted(kw.items())), B, Tn, seed, ragged)
    if key in _CACHE:
        return _CACHE[key]
    from oracle import radmmm_oracle as O
    cfg = O.DecoderConfig(**kw)
    sd = _T(O.procedural_decoder_state(O.decoder_state_shapes(cfg)))
    b = _T(O.synthetic_batch(B, Tn, cfg, seed, ragged=ragged))
    p = {k: (v.clone().requires_grad_(True) if v.dtype == torch.float32 and "running" not in k
             and not k.endswith((".p", "lower_diag", "input_mean")) else v) for k, v in sd.items()}
    omel = b["mel"].clone().requires_grad_(True)
    octx = b["context"].clone().requires_grad_(True)
    ro = O.decoder_forward(p, cfg, omel, b["spk"], octx, b["lengths"], b["f0"], b["energy"], b["accent"])
    lo, _ = O.decoder_loss(ro, b["lengths"], cfg.n_group_size)
    lo.backward()
    ul = b["lengths"] // cfg.n_group_size
    m = (torch.arange(Tn // cfg.n_group_size)[None] < ul[:, None])[:, None]
    res = {"cfg": cfg, "sd": sd, "batch": b, "mask": m, "z_mel": ro["z_mel"].detach(),
           "log_det_W_list": [float(x) for x in ro["log_det_W_list"]],
           "log_s_sums": [float((c.detach() * m).sum()) for c in ro["log_s_list"]],
           "loss": float(lo.detach()), "grads": {k: v.grad for k, v in p.items() if torch.is_tensor(v) and v.requires_grad and v.grad is not None},
           "g_mel": omel.grad, "g_ctx": octx.grad}
    _CACHE[key] = res
    return res


def drop(kw, B, Tn, seed, ragged=True):
    """Release a cached run (the T = 2000 one holds ~2 GB)."""
    _CACHE.pop((tuple(sorted(kw.items())), B, Tn, seed, ragged), None)
