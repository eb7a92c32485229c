"""Session cache of full-size CPU-oracle runs (forward + NLL + whole backward of the decoder).  Several GPU tests compare
different product schemes / kernels against the oracle on THE SAME seeded batch and weights; the oracle result depends on
neither, so it is computed once per (config, batch shape, seed) and shared.  Test infrastructure only."""
import numpy as np
import torch

_CACHE = {}


def _T(d):
    return {k: torch.from_numpy(np.asarray(v)) for k, v in d.items()}


def oracle_decoder_run(kw, B, Tn, seed, ragged=True, knot_ulps=None, frame_weight_fn=None):
    """{"sd", "batch", "z_mel", "log_det_W_list", "log_s_sums", "loss", "grads" (per parameter name), "g_mel", "g_ctx"} for
    the procedural weights of DecoderConfig(**kw) on synthetic_batch(B, Tn, seed, ragged).  Tensors are detached.
    knot_ulps (configs with spline flows): the frames that hold a spline element within that many fp32 ulp of a bin edge
    are LOCATED in the forward pass and taken out of the loss the backward starts from ("frame_weight" [B, 1, T'], 0 at
    those frames; "knot_elements" / "knot_frames": the counts; "loss" stays the unweighted NLL) -- the log-Jacobian's
    gradient has a kink at every knot, so another implementation's gradient is only comparable away from them.
    frame_weight_fn(recs, mask) -> bool [B*T'] of further frames to exclude: called between the oracle's forward and its
    backward with the per-flow records ("bins": the bin every element's search picked, "edge_dist", "near"), so that a test
    can run the OTHER implementation's forward there and hand back the frames whose bin decisions differ."""
    key = (tuple(sorted(kw.items())), B, Tn, seed, ragged, knot_ulps, frame_weight_fn is not None)
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
    recs = [{"ulps": knot_ulps} for _ in range(cfg.n_splines)] if (knot_ulps and cfg.n_splines) else None
    ro = O.decoder_forward(p, cfg, omel, b["spk"], octx, b["lengths"], b["f0"], b["energy"], b["accent"], spline_records=recs)
    lo, _ = O.decoder_loss(ro, b["lengths"], cfg.n_group_size)
    ul = b["lengths"] // cfg.n_group_size
    m = (torch.arange(Tn // cfg.n_group_size)[None] < ul[:, None])[:, None]
    extra = {}
    if recs is not None:
        Tg = Tn // cfg.n_group_size
        near = torch.zeros(B * Tg, dtype=torch.bool)
        n_el = 0
        for r in recs:
            nr = r["near"] & m.reshape(-1, 1)
            n_el += int(nr.sum())
            near |= nr.any(1)
        if frame_weight_fn is not None:
            more = frame_weight_fn(recs, m)
            extra["flipped_frames"] = int((more & ~near).sum())
            near |= more
        w = (~near).reshape(B, 1, Tg).float()
        n_elem = torch.div(b["lengths"].sum(), cfg.n_group_size, rounding_mode="floor")
        lw, _ = O.compute_flow_loss(ro["z_mel"], ro["log_det_W_list"], ro["log_s_list"], n_elem, ro["z_mel"].shape[1], m.float() * w)
        lw.backward()
        extra.update({"leaky_near_zero": sum(r.get("leaky_near_zero", 0) for r in recs),
                      "leaky_total": sum(r.get("leaky_total", 0) for r in recs), "frame_weight": w, "knot_elements": n_el, "knot_frames": int(near.sum()), "weighted_loss": float(lw.detach())})
    else:
        lo.backward()
    res = {"cfg": cfg, "sd": sd, "batch": b, "mask": m, "z_mel": ro["z_mel"].detach(), **extra,
           "log_det_W_list": [float(x) for x in ro["log_det_W_list"]],
           "log_s_sums": [float((c.detach() * m).sum()) for c in ro["log_s_list"]],
           "loss": float(lo.detach()), "grads": {k: v.grad for k, v in p.items() if torch.is_tensor(v) and v.requires_grad and v.grad is not None},
           "g_mel": omel.grad, "g_ctx": octx.grad}
    _CACHE[key] = res
    return res


def drop(kw, B, Tn, seed, ragged=True, knot_ulps=None, with_fn=False):
    """Release a cached run (the T = 2000 one holds ~2 GB)."""
    _CACHE.pop((tuple(sorted(kw.items())), B, Tn, seed, ragged, knot_ulps, with_fn), None)
