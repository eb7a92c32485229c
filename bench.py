#!/usr/bin/env python3
"""Headline benchmark: mel-frames/s of the RADTTS flow-decoder training step
(decoder forward + flow NLL + backward; no optimizer, no data loading) on synthetic
fixed-length batches, B=32 per GPU, 80 mel, T=800 (BASELINE.json configs[1]).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

One process per GPU; N > 1 adds the bucketed RCCL gradient all-reduce overlapped with
backward (rad_mmm_amd/ddp.py).  Rank 0 prints ONE JSON line.  Extra objects:
  roofline     dominant kernel (dilated k=5 1024->1024 conv GEMM on the split-operand path: f16 hi.hi product + FP8
               cross terms by default) timed live with HIP events on the launch stream: algorithmic FLOP per launch /
               avg duration against the dense f16 MFMA peak; traffic from profiles/pmc_dominant.json (static)
  exact_split_mode / throughput_mode   the same step with three f16 products / one f16 product (medians of 5 steps)
  wn_stack     the north_star's derived view: algorithmic fp32 bytes of the WN stack / step time
  cpu_baseline the CPU oracle (a port of the reference's arithmetic) timed on the host cores
               on a bounded sample of the same workload (rank 0, N=1 only)
Other workloads beside the headline line: --full-step (the whole training step of the reference's loop), --config joint
(BASELINE configs[3]: RADMMM decoder + the four attribute predictors in ONE step, with its own parity block against the CPU
oracle's restatement of TTSModel.training_step), --config radmmm / radmmm_splines (configs[2] / configs[4]).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

RADTTS = dict(n_speaker_dim=16, use_accent_emb_for_decoder=True, n_accent_dim=8, n_text_dim=512, n_f0_dims=1,
              n_energy_avg_dims=1, n_mel_channels=80, n_early_size=2, n_early_every=2, n_group_size=2,
              scaling_fn="tanh", affine_activation="softplus", use_partial_padding=True,
              n_conv_layers_per_step=4, n_flows=8)

# BASELINE configs[4]: 16 kHz / RADMMM dims (n_text_dim 520, accent not in decoder) with 2 spline steps
RADMMM_SPLINES = dict(RADTTS, n_text_dim=520, use_accent_emb_for_decoder=False, n_splines=2, use_bn=True)
# BASELINE configs[2]: the shipped RADMMM decoder (configs/RADMMM_model_config.yaml:16-39): 8 affine flows, D = 1056
RADMMM = dict(RADTTS, n_text_dim=520, use_accent_emb_for_decoder=False)
# "joint" = BASELINE configs[3]: the RADMMM decoder + the four attribute predictors in one training step (full_step leg)
CONFIGS = {"radtts": RADTTS, "radmmm": RADMMM, "radmmm_splines": RADMMM_SPLINES, "joint": RADMMM}

PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_F16_MFMA_TFLOPS = 2500.0   # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_f16, dense (no sparsity)
PEAK_HBM_GBS = 8000.0


def procedural_state(cfg_kwargs):
    """Random-init weights of the named architecture (no checkpoints exist offline): the same
    closed-form generator the golden fixtures use, so every rank builds identical weights."""
    import radmmm_synth as O
    cfg = O.DecoderConfig(**cfg_kwargs)
    sd = O.procedural_decoder_state(O.decoder_state_shapes(cfg))
    return cfg, {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}


def algorithmic_flops_per_frame(cfg) -> float:
    """fwd+bwd FLOPs per MEL frame of the conv/1x1 stack (bwd = 2x fwd)."""
    D, W, L = cfg.cond_dims, 1024, cfg.n_conv_layers_per_step
    fwd = 0.0
    for C in cfg.flow_channels():
        h = C // 2
        fwd += 2 * (C * C + (h + D) * W + L * (5 * W * W + W * W) + W * C)
    return 3.0 * fwd / cfg.n_group_size


def algorithmic_bytes_per_frame(cfg) -> float:
    """fp32 HBM bytes per mel frame fwd+bwd under one-kernel-per-conv fusion."""
    D, W, L = cfg.cond_dims, 1024, cfg.n_conv_layers_per_step
    fl = 0.0
    for C in cfg.flow_channels():
        h = C // 2
        fl += 2 * C + (h + D + W) + L * 2 * W * 2 + (3 * L * W - W) + (W + 3 * C // 2)
    return 3.0 * 4.0 * fl / cfg.n_group_size


def time_dominant_kernel(N, T, reps=20):
    """Average duration (s) of ONE launch of the dominant kernel: rowgemm_f32 as the WN in_layer
    forward conv (M=B*T', N=1024, K=5x1024, dilation 2, partial-conv epilogue + softplus),
    bracketed by HIP events on the stream the kernel is launched on (torch's current stream)."""
    from rad_mmm_amd._lib import rowgemm
    dev = torch.device("cuda", torch.cuda.current_device())
    g = torch.Generator(device="cpu").manual_seed(0)
    x = torch.randn(N, 1024, generator=g).to(dev)
    w = (torch.randn(5, 1024, 1024, generator=g) * 0.02).to(dev)
    b = torch.zeros(1024, device=dev)
    y = torch.empty(N, 1024, device=dev)
    lens = torch.full((N // T,), T, dtype=torch.int32, device=dev)

    def launch():
        rowgemm(A=x, lda=1024, B=w, ldb=1024, b_tap_stride=1024 * 1024, b_layout=0, C=y, ldc=1024, M=N, N=1024,
                K=1024, taps=5, dil=2, sign=1, T=T, lens=lens, a_mask_mode=1, bias=b, pconv=1, ratio_taps=5,
                ratio_dil=2, postmask=1, act=1)
    for _ in range(3):
        launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        launch()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps, 2.0 * N * 1024 * 5 * 1024


def time_dominant_kernel_h3(N, T, reps=20, nprod=3):
    """Same launch as time_dominant_kernel on the split-operand path (nprod 3: split-f16, 2: f16 + FP8 cross terms):
    rowgemm_h3 as the WN in_layer forward conv, split activations in, fp32 + split copies out."""
    from rad_mmm_amd._lib import rowgemm_h3
    from rad_mmm_amd import ops
    dev = torch.device("cuda", torch.cuda.current_device())
    g = torch.Generator(device="cpu").manual_seed(0)
    x = torch.nn.functional.softplus(torch.randn(N, 1024, generator=g)).to(dev)
    v = (torch.randn(1024, 1024, 5, generator=g) * 0.02).to(dev)
    gg = torch.ones(1024, 1, 1, device=dev)
    b = torch.zeros(1024, device=dev)
    xh, xl = ops.split_f16(x, 1024, 1.0, 1024, nprod, ops.X8_ACT_EXP)
    Wh, Wl, _ = ops.split_weight(v, gg, 1024, nprod=nprod)
    y = torch.empty(N, 1024, device=dev)
    yh, yl = torch.empty_like(xh), torch.empty_like(xl)
    lens = torch.full((N // T,), T, dtype=torch.int32, device=dev)
    # the launch writes what the step's launch writes: since round 5 the FP8-cross scheme keeps the hidden state as its split
    # pair only (no fp32 copy: C = NULL; ops.AffineFlowStepH3Fn `pair_only`)
    keep_c = not (nprod == 2 and os.environ.get("RADMMM_KEEP_FP32", "0") != "1")

    def launch():
        rowgemm_h3(nprod=nprod, a8_exp=ops.X8_ACT_EXP, b8_exp=ops.X8_W_EXP, split_fmt=ops.fmt_a(nprod), ch_x8_exp=ops.X8_ACT_EXP,
                   Ah=xh, Al=xl, lda_h=1024, Bh=Wh, Bl=Wl, ldb_h=1024, b_tap_stride_h=Wh.stride(0),
                   acc_scale=1.0 / ops.W_SCALE, C=y if keep_c else None, ldc=1024, M=N, N=1024, K=1024, taps=5, dil=2, sign=1, T=T,
                   lens=lens, a_mask_mode=1, bias=b, pconv=1, ratio_taps=5, ratio_dil=2, postmask=1, act=1, Ch=yh, Cl=yl, ldch=1024,
                   ch_scale=1.0)
    for _ in range(3):
        launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        launch()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps, 2.0 * N * 1024 * 5 * 1024


def time_wgrad_h3(N, T, reps=10):
    """in_layer weight gradient on the split-f16 path incl. the two transposing split producers."""
    from rad_mmm_amd import ops
    dev = torch.device("cuda", torch.cuda.current_device())
    g = torch.Generator(device="cpu").manual_seed(1)
    gy = torch.randn(N, 1024, generator=g).to(dev)
    x = torch.randn(N, 1024, generator=g).to(dev)
    B = N // T
    lens = torch.full((B,), T, dtype=torch.int32, device=dev)

    def run(with_producers=True):
        gy_t = ops.transpose_split_act(gy, 1024, B, T, None, 0, 1.0, "gy")
        x_t = ops.transpose_split_act(x, 1024, B, T, lens, 1, 1.0, "x")
        return gy_t, x_t
    gy_t, x_t = run()
    ops.wgrad_h3_slabs(gy_t, x_t, 1024, 1024, 1024, 5, 2, 1.0)
    out = []
    for what in ("gemm", "gemm+producers"):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            if what != "gemm":
                gy_t, x_t = run()
            ops.wgrad_h3_slabs(gy_t, x_t, 1024, 1024, 1024, 5, 2, 1.0)
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) * 1e-3 / reps)
    return out, 2.0 * N * 1024 * 5 * 1024


def time_wgrad_kernel(N, T, reps=10):
    """Average duration of the in_layer weight-gradient launch (5 taps, split-K as the step uses)."""
    from rad_mmm_amd import ops
    dev = torch.device("cuda", torch.cuda.current_device())
    g = torch.Generator(device="cpu").manual_seed(1)
    gy = torch.randn(N, 1024, generator=g).to(dev)
    x = torch.randn(N, 1024, generator=g).to(dev)
    lens = torch.full((N // T,), T, dtype=torch.int32, device=dev)
    for _ in range(2):
        ops.wgrad_slabs(gy, 1024, x, 1024, 1024, T, lens, taps=5, dil=2, x_mask_mode=1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        ops.wgrad_slabs(gy, 1024, x, 1024, 1024, T, lens, taps=5, dil=2, x_mask_mode=1)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps, 2.0 * N * 1024 * 5 * 1024


def host_threads() -> int:
    """CPU threads this process may really use: affinity mask capped by the cgroup CPU quota
    (os.cpu_count() reports the whole host and oversubscribing it is catastrophically slow)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return max(1, min(n, 32))


def _latest_trace(suffix):
    """profiles/r0N_<suffix> of the latest round that committed one (the kernel traces quoted by roofline_hbm / roofline_mfma)"""
    for r in (6, 5):
        name = f"r0{r}_{suffix}"
        if os.path.exists(os.path.join(ROOT, "profiles", name)):
            return name
    return f"r05_{suffix}"


def mfma_rooflines(dec, cfg, B, T):
    """EVERY GEMM class of the headline step against the dense f16 MFMA peak, not only the best one
    (`roofline` = the 5-tap forward).  Algorithmic flops per step (2 M N K taps per launch) are computed here from
    the model's shapes; the in-step durations are per-kernel sums over a step and are QUOTED from the committed rocprofv3
    kernel trace of this very workload (tools/prof_step.sh), labelled static, exactly as `roofline_hbm` does.  Launches that
    share a kernel name (one template instantiation serving two convs) are one row with the flops of both."""
    n_spl = sum(1 for f in dec.flows if getattr(f, "use_spline", False))
    if (B, T) != (32, 800) or n_spl or cfg.cond_dims != 1048 or dec.gemm_precision != "f8x":
        return None
    trace = _latest_trace("kernel_stats.json")
    try:
        with open(os.path.join(ROOT, "profiles", trace)) as f:
            ks = json.load(f)["kernels"]
    except Exception:
        return None
    M = B * (T // cfg.n_group_size)
    nf = len(dec.flows)
    wn = dec.flows[0].coupling_tfn.affine_param_predictor
    W = wn.n_channels
    nl = wn.n_layers
    Kp = (wn.start.weight_v.shape[1] + 31) // 32 * 32
    C = wn.end.weight.shape[0]
    g = lambda n, k, taps=1: 2.0 * M * n * k * taps
    rows = [
        # (name substring in the trace, what, launches per step, flop per step)
        ("rowgemm_win_kernelILi7ELi2ELb0", "in_layer forward (5 taps, softplus, split pair) x %d + in_layer 0's data gradient" % nl,
         nf * (nl + 1), nf * (nl + 1) * g(W, W, 5)),
        ("rowgemm_win_kernelILi7ELi4ELb1", "fused data gradient: 5 taps of in_layer j+1 + the 1x1 of res_skip j as one launch",
         nf * (nl - 1), nf * (nl - 1) * (g(W, W, 5) + g(W, W))),
        ("wgrad_rm8_kernel", "all weight gradients of the WN convs (4 x 5 taps, 4 x 1 tap, start, end per flow step)",
         nf * (2 * nl + 2), nf * (nl * g(W, W, 5) + nl * g(W, W) + g(W, Kp) + g(C, W))),
        ("rowgemm_one_kernelILi7ELi1E", "res_skip forward, layers 0..%d (1x1, softplus, fp32 out)" % (nl - 2), nf * (nl - 1), nf * (nl - 1) * g(W, W)),
        ("rowgemm_one_kernelILi7ELi3E", "res_skip forward, last layer (adds the earlier outputs, writes the skip sum's pair)", nf, nf * g(W, W)),
        ("rowgemm_one_kernelILi7ELi4E", "res_skip data gradient of the last layer (1x1)", nf, nf * g(W, W)),
        ("rowgemm_one_kernelILi7ELi2E", "start conv forward (1x1, K = %d, split pair)" % Kp, nf, nf * g(W, Kp)),
        ("rowgemm_one_kernelILi8ELi1E", "start conv data gradient (1x1, N = %d)" % Kp, nf, nf * g(Kp, W)),
        ("rowgemm_one_kernelILi4ELi1E", "end conv forward (1x1, N = %d)" % C, nf, nf * g(C, W)),
    ]
    out = []
    for sub, what, launches, flop in rows:
        hit = [v for k, v in ks.items() if sub in k and v.get("ms_per_step")]
        ms = sum(v["ms_per_step"] for v in hit)
        if ms <= 0:
            continue
        tf = flop / (ms * 1e-3) / 1e12
        out.append({"kernel": sub, "what": what, "launches_per_step": launches, "ms_per_step": ms, "flop_per_step": flop,
                    "achieved": tf, "peak": PEAK_F16_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": tf / PEAK_F16_MFMA_TFLOPS})
    one = [r for r in out if r["kernel"].startswith("rowgemm_one")]
    return {"bound": "mfma", "durations_static": True,
            "durations_source": f"profiles/{trace} (rocprofv3 --kernel-trace over `bench.py --step-only` of this workload, tools/prof_step.sh)",
            "note": "algorithmic flops (one fp32-class product per multiply-add) against the dense f16 MFMA peak; the FP8-cross scheme "
                    "executes two f16-equivalent MFMA passes per product, so the matrix pipe's own utilisation is twice `frac`",
            "one_tap_family_ms_per_step": sum(r["ms_per_step"] for r in one),
            "kernels": out}


def hbm_rooflines(dec, cfg, B, T):
    """SURVEY 8(d)'s second regime: the HBM-bound kernel classes of the step, each as algorithmic bytes per step / in-step time
    per step against the 8 TB/s peak.  Bytes are computed here from the model's shapes; the in-step durations cannot be taken
    from inside the process (they are per-kernel sums over a step) and are QUOTED from the committed rocprofv3 kernel trace of
    this very workload, profiles/r0N_kernel_stats.json of the latest round (tools/prof_step.sh), labelled static."""
    import math
    # a trace belongs to ONE workload: the default bench line's (RADTTS, B = 32, T = 800) or BASELINE configs[4]'s
    # (`--config radmmm_splines --frames 2000`); any other shape has no committed trace and gets no static figures
    n_spl = sum(1 for f in dec.flows if getattr(f, "use_spline", False))
    if (B, T) == (32, 800) and not n_spl and cfg.cond_dims == 1048:
        trace = _latest_trace("kernel_stats.json")
    elif (B, T) == (32, 2000) and n_spl == 2:
        trace = _latest_trace("c5_kernel_stats.json")
    else:
        return None
    path = os.path.join(ROOT, "profiles", trace)
    try:
        with open(path) as f:
            ks = json.load(f)["kernels"]
    except Exception:
        return None
    from rad_mmm_amd import ops
    N = B * (T // cfg.n_group_size)
    slots = int(ops.lib.radmmm_gemm_cu_slots())
    wn_elems, bwd_bytes = 0, 0.0
    tr_bytes = 0.0
    for f in dec.flows:
        if getattr(f, "use_spline", False):
            continue
        wn = f.coupling_tfn.affine_param_predictor
        convs = [wn.start.weight_v] + [l.conv.weight_v for l in wn.in_layers] + [r.weight_v for r in wn.res_skip_layers]
        for v in convs:
            co, ci, k = v.shape
            n = co * round(math.ceil(ci / 32) * 32) * k
            wn_elems += n
            tiles = math.ceil(co / 256) * math.ceil(ci / 256) * k
            S = ops.pick_splits(tiles, N, slots=slots)
            bwd_bytes += n * 4.0 * (S + 2)              # S split-K slabs + v read, g_v written
            tr_bytes += n * 4.0 * 2                      # split pair read, transposed pair written
        wn_elems += wn.end.weight.numel()
    C = 160
    rows = {
        "weightnorm_fwd_h3": ("weight norm + scale + split of every conv weight, one launch per flow step (fp32 v read, fp16 hi + 8-bit cross written)", wn_elems * 8.0),
        "weightnorm_bwd": ("weight-norm backward over the split-K slabs of the weight gradients", bwd_bytes),
        "transpose_pair_x8": ("transposed copies of the split weights, one launch per flow step for the data-gradient GEMMs", tr_bytes),
        "dact_rows_multi_kernel": ("gQ_j = gOUT * softplus'(R_j), j < 4, in one pass per flow step: gOUT read once, four fp32 R read, four split pairs written",
                                   N * 1024 * (4 + 4 * (4 + 4.0)) * (len(dec.flows) - n_spl)),
        "affine_coupling_fwd_kernel": ("affine coupling forward (O, z1 read; z, log s written)", N * (C + C + C + C / 2) * 4.0 * (len(dec.flows) - n_spl)),
        "affine_coupling_bwd_kernel": ("affine coupling backward", N * (C * 5 + C / 2) * 4.0 * (len(dec.flows) - n_spl)),
        "wn_input_fwd4_kernel": ("WN input assembly [context | z half] written as its split pair only (round 5: no fp32 copy)",
                                 N * (1048 + 80 + 1152) * 4.0 * (len(dec.flows) - n_spl)),
    }
    out = []
    for sub, (what, bytes_step) in rows.items():
        ms = sum(v["ms_per_step"] for k, v in ks.items() if sub in k and v.get("ms_per_step"))
        if ms <= 0:
            continue
        gbs = bytes_step / (ms * 1e-3) / 1e9
        out.append({"kernel": sub, "what": what, "algorithmic_gb_per_step": bytes_step / 1e9, "ms_per_step": ms,
                    "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS})
    res = {"bound": "hbm", "durations_static": True,
           "durations_source": f"profiles/{trace} (rocprofv3 --kernel-trace over `bench.py --step-only` of this workload, tools/prof_step.sh)",
           "kernels": out}
    if n_spl:
        # BASELINE configs[4]: the piecewise-quadratic transform, HBM-bound on the predicted parameters q (2K+1 = 65 floats
        # per element each way)
        k5 = ks
        h, K = dec.flows[0].coupling_tfn.half_mel_channels, dec.flows[0].coupling_tfn.K
        qb = N * h * (2 * K + 1) * 4.0
        for sub, what, byt in (("pq_spline_fwd", "spline forward: q read; x read, y + log-Jacobian written", qb + N * h * 12.0),
                               ("pq_spline_bwd", "spline backward: q read, dq written; x, gy read, gx written", 2 * qb + N * h * 12.0)):
            ms = sum(v["ms_per_step"] for k, v in k5.items() if sub in k and v.get("ms_per_step"))
            if ms > 0:
                gbs = byt * n_spl / (ms * 1e-3) / 1e9
                out.append({"kernel": sub, "what": what, "algorithmic_gb_per_step": byt * n_spl / 1e9, "ms_per_step": ms,
                            "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS})
    return res


def cpu_baseline(cfg, sd, batch, hip_out, budget_s=25.0):
    """The CPU oracle (fp32 restatement of the reference, kind="port") timed on the host cores on a bounded sample of THE
    SAME workload: the first Bs utterances of the bench batch (fixed length: the utterances are independent given the
    weights), Bs the largest of 32, 16, 8, 4, 2, 1 whose predicted time fits the budget (prediction: Bs x the time of the
    batch's first utterance alone, an upper bound).  Because the sample is part of the batch the HIP step ran on, the same call yields the
    run's own parity figures: z of those utterances (max abs difference / max abs value) and their NLL, from `hip_out` =
    the decoder's outputs on the full batch, against the oracle's (reference loss.py:85-110)."""
    from oracle import radmmm_oracle as O
    threads = host_threads()
    torch.set_num_threads(threads)
    p = {k: (v.clone().requires_grad_(True) if v.dtype == torch.float32 and v.dim() > 0
             and not k.endswith((".p", "lower_diag", "input_mean")) else v) for k, v in sd.items()}
    full = {k: torch.from_numpy(v) for k, v in batch.items()}

    def one(b):
        for v in p.values():
            if v.requires_grad:
                v.grad = None
        t0 = time.perf_counter()
        out = O.decoder_forward(p, cfg, b["mel"], b["spk"], b["context"], b["lengths"], b["f0"], b["energy"], b["accent"])
        lm, _ = O.decoder_loss(out, b["lengths"], cfg.n_group_size)
        lm.backward()
        return time.perf_counter() - t0, float(lm.detach()), out["z_mel"].detach()

    Bt, T = full["mel"].shape[0], full["mel"].shape[-1]
    first = {k: v[:1] for k, v in full.items()}
    one(first)                                    # warm-up (thread pools, oneDNN primitives)
    probe, _, _ = one(first)                      # one utterance of the batch: an upper bound of the cost per utterance
    Bs = 1
    for cand in (32, 16, 8, 4, 2):
        if cand <= Bt and probe * cand <= budget_s:
            Bs = cand
            break
    sub = {k: v[:Bs] for k, v in full.items()}
    dt, lo, zo = one(sub)
    # the HIP step's outputs on the same utterances: z, and their NLL in closed form (fixed length: every frame counts)
    g = cfg.n_group_size
    zh = hip_out["z_mel"][:Bs].detach().float().cpu()
    Tg = zh.shape[-1]
    nll_h = (float((zh.double() ** 2).sum()) / 2 - sum(float(ls[:Bs].double().sum()) for ls in hip_out["log_s_list"])
             - float(torch.stack(list(hip_out["log_det_W_list"])).double().sum()) * Tg * Bs) / (Bs * Tg * zh.shape[1])
    parity = {"utterances": Bs, "frames": T,
              "z_rel_err_vs_cpu": float((zh - zo).abs().max() / zo.abs().max()),
              "nll_hip": nll_h, "nll_cpu": lo, "nll_abs_diff_vs_cpu": abs(nll_h - lo), "nll_rel_diff_vs_cpu": abs(nll_h - lo) / abs(lo),
              "bar": "1e-4 relative (BASELINE.json north_star)"}
    return {"value": Bs * T / dt, "unit": "mel-frames/s", "cores": threads, "kind": "port",
            "sample": f"oracle decoder fwd+NLL+bwd on the first {Bs} utterances of the bench batch (8 flows WN-1024, T={T}, fp32), "
                      f"1 step = {dt:.1f} s (one utterance alone: {probe:.2f} s)"}, parity


# BASELINE configs[3]: the four attribute predictors of the joint step at the dims their YAMLs ship
# (configs/RADMMM_{f0,energy,vpred,duration}model_config.yaml: ConvLSTMLinearDAP, in_dim 520, reduction 16, 3 conv layers of
# 256 channels, kernel 5, dropout 0.5, accent embedding on; all four with loss.AttributeRegressionLoss, weight 1)
JOINT_PREDICTORS = {"f0": dict(target_offset=-5.0, prefix="f0_"), "energy": dict(target_offset=-0.75, prefix="energy_"),
                    "voiced": dict(prefix="vpred_"), "duration": dict(log_target=True, prefix="duration_")}


def build_joint_predictors(CFG, p_dropout=0.5):
    from rad_mmm_amd.attribute_predictors import AttributeRegressionLoss, ConvLSTMLinearDAP
    out = {}
    for name, spec in JOINT_PREDICTORS.items():
        out[f"{name}_predictor"] = ConvLSTMLinearDAP(
            n_speaker_dim=CFG["n_speaker_dim"], n_accent_dim=CFG["n_accent_dim"], use_accent_embedding=True, in_dim=CFG["n_text_dim"],
            out_dim=1, reduction_factor=16, n_backbone_layers=3, n_hidden=256, kernel_size=5, p_dropout=p_dropout,
            target_offset=spec.get("target_offset", 0.0), log_target=spec.get("log_target", False), lstm_type="bilstm")
        out[f"{name}_predictor_loss"] = AttributeRegressionLoss(spec["prefix"], 1.0)
    return out


def build_step_model(dec, CFG, dev, joint=False):
    """TTSTrainingStep around `dec` as the full-step / joint legs run it (text encoder, attention, criterion; joint: + the four
    attribute predictors of configs[3] with their spectral norms' power iteration converged)"""
    from rad_mmm_amd.encoder import Encoder
    from rad_mmm_amd.loss import RADMMMLoss
    from rad_mmm_amd.tts_step import TTSTrainingStep
    torch.manual_seed(1234)
    extra = build_joint_predictors(CFG) if joint else {}
    model = TTSTrainingStep(Encoder(3, CFG["n_text_dim"], 5), dec, RADMMMLoss(sigma=1.0, kl_loss_start_iter=0),
                            n_speakers=8, n_accents=4, n_text_tokens=185, n_text_dim=CFG["n_text_dim"],
                            n_speaker_dim=CFG["n_speaker_dim"], n_accent_dim=CFG["n_accent_dim"], use_accent=True,
                            use_accent_emb_for_decoder=CFG["use_accent_emb_for_decoder"], binarization_start_iter=0,
                            **extra).to(dev).train()
    if joint:
        for name in JOINT_PREDICTORS:                       # converge spectral norm's power iteration (fresh u / v leave |W_hh| ~ 10)
            lstm = getattr(model, f"{name}_predictor").feat_pred_fn.bilstm
            for _ in range(20):
                for hook in lstm._forward_pre_hooks.values():
                    hook(lstm, ())
    return model


def build_step_batch(gb, B, T, dev, t_txt=150, joint=False):
    """synthetic text (t_txt tokens per utterance) around the decoder leg's mel / f0 / energy, with the host copies of the
    lengths the collate function has anyway; joint: + a voiced mask"""
    from rad_mmm_amd.data import BetaBinomialInterpolator
    g = torch.Generator().manual_seed(99)
    in_lens = [t_txt] * B
    batch = {"mel": gb["mel"] * 2 - 5,                      # the step applies (mel + 5) / 2 itself
             "speaker_ids": torch.randint(0, 8, (B,), generator=g).to(dev), "accent_ids": torch.randint(0, 4, (B,), generator=g).to(dev),
             "text": torch.randint(0, 185, (B, t_txt), generator=g).to(dev),
             "input_lengths": torch.tensor(in_lens, device=dev), "output_lengths": gb["lengths"],
             "input_lengths_host": torch.tensor(in_lens), "output_lengths_host": gb["lengths"].cpu(),
             "attn_prior": BetaBinomialInterpolator(device=dev).batch(in_lens, [T] * B),
             "f0": gb["f0"], "energy_avg": gb["energy"]}
    if joint:
        batch["voiced_mask"] = (gb["f0"] > gb["f0"].median()).float()      # synthetic: the upper half of the f0 track counts as voiced
    return batch


def joint_parity_vs_cpu(model, batch, cfg, Bs, dev):
    """The joint step on the first Bs utterances of the bench batch, HIP against the CPU oracle's restatement of
    TTSModel.training_step (oracle.tts_joint_step), both WITHOUT dropout (the reference's only random element): the summed
    loss, every loss term, the four predictors' outputs and the decoder's z.  -> (report dict, worst relative figures)"""
    import torch.nn.functional as F
    from oracle import radmmm_oracle as O
    from rad_mmm_amd.common import SequenceLength
    nb = batch["mel"].shape[0]
    sub = {k: (v[:Bs] if torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == nb else v) for k, v in batch.items()}
    real_dropout = F.dropout
    F.dropout = lambda x, p=0.5, training=True, inplace=False: x
    try:
        with torch.no_grad():
            loss, losses, outs = model.training_step(sub, global_step=10)
            ctx_d, spk_d, acc_d = outs["context"], outs["spk_vecs"], outs["accent_vecs"]
            ol = SequenceLength(sub["output_lengths"], sub.get("output_lengths_host"))
            il = SequenceLength(sub["input_lengths"], sub.get("input_lengths_host"))
            txt_enc, _ = model.encode_text(sub["text"], il.lengths, None, int(il.lengths_host.max()))
            pred = {"f0": model.f0_predictor(sub["f0"].unsqueeze(1), ctx_d, spk_d, ol, None, None, acc_d)["x_hat"],
                    "energy": model.energy_predictor(sub["energy_avg"].unsqueeze(1), ctx_d, spk_d, ol, accent_emb=acc_d)["x_hat"],
                    "voiced": model.voiced_predictor(sub["voiced_mask"].unsqueeze(1), ctx_d, spk_d, ol, accent_emb=acc_d)["x_hat"],
                    "duration": model.duration_predictor(outs["attn"].sum(2), txt_enc, spk_d, il, accent_emb=acc_d)["x_hat"]}
    finally:
        F.dropout = real_dropout
    torch.cuda.synchronize()
    p = {n: v.detach().float().cpu() if v.is_floating_point() else v.detach().cpu() for n, v in model.state_dict().items()}
    cb = {k: (v.detach().cpu() if torch.is_tensor(v) else v) for k, v in sub.items()}
    specs = {name: dict(n_layers=3, weight=1.0, **spec) for name, spec in JOINT_PREDICTORS.items()}
    t0 = time.perf_counter()
    with torch.no_grad():
        ref = O.tts_joint_step(p, cfg, cb, specs, binarize=True, bin_loss=True)
    dt = time.perf_counter() - t0
    rel = lambda a, b: abs(float(a) - float(b)) / max(abs(float(b)), 1e-12)
    terms = {k: {"hip": float(v[0]), "cpu": float(ref["losses"][k][0]), "rel_diff": rel(v[0], ref["losses"][k][0])}
             for k, v in losses.items() if k in ref["losses"] and torch.is_tensor(v[0])}
    z_err = float((outs["z_mel"].cpu() - ref["z_mel"]).abs().max() / ref["z_mel"].abs().max())
    perr = {}
    for name, xh in pred.items():
        r = ref["pred"][name]
        w = r.shape[2]
        lens = cb["input_lengths"] if name == "duration" else cb["output_lengths"]
        m = (torch.arange(w)[None, :] < lens[:, None])[:, None]
        perr[name] = float(((xh.cpu()[:, :, :w] - r) * m).abs().max() / (r * m).abs().max())
    hard_h, hard_c = outs["attn"].detach().cpu().round(), ref["attn"].round()
    attn_same = int((hard_h == hard_c).flatten(1).all(1).sum())
    rep = {"utterances": Bs, "dropout": "off on both sides (the step's only random element)", "oracle_seconds": round(dt, 1),
           "summed_loss_hip": float(loss), "summed_loss_cpu": float(ref["loss"]), "summed_loss_rel_diff": rel(loss, ref["loss"]),
           "loss_terms": terms, "z_rel_err_vs_cpu": z_err, "predictor_output_rel_err_vs_cpu": perr,
           "hard_alignments_identical": f"{attn_same} of {Bs}",
           "bar": "1e-4 relative on the outputs (BASELINE.json north_star); alignments bit-exact on equal logs"}
    worst = {"loss": rep["summed_loss_rel_diff"], "terms": max(t["rel_diff"] for t in terms.values()), "z": z_err,
             "pred": max(perr.values()), "alignments_identical": attn_same}
    return rep, worst


def full_step_leg(dec, cfg, CFG, gb, B, T, dev, decoder_only_ms, steps=7, t_txt=150, joint=False, parity_utts=0):
    """The WHOLE training step the reference's Lightning loop runs per batch (tts_lightning_modules.py:643-750 + clip +
    RAdam, configs/RADMMM_train_config.yaml:7-8): embeddings, text encoder, alignment attention with the beta-binomial
    prior, on-device MAS (binarisation on), context = txt_enc . attn^T, flow decoder, flow NLL + CTC + binarisation losses,
    backward, global-norm clip 1.0, RAdam -- on synthetic text (150 tokens per utterance) and the decoder leg's mel.
    Gradients land in the flat buckets of a BucketedGradReducer around the whole step (what a data-parallel run uses) and
    FlatRAdam steps on those buckets; the batch carries the host copies of the lengths the collate function has anyway.
    Reported: ms per step, the share outside the decoder's fwd+bwd, and the host synchronisations one step makes."""
    import warnings
    from rad_mmm_amd.ddp import BucketedGradReducer
    from rad_mmm_amd.optim import FlatRAdam
    model = build_step_model(dec, CFG, dev, joint)
    batch = build_step_batch(gb, B, T, dev, t_txt, joint)
    joint_parity = None
    if joint and parity_utts > 0:                           # before the first optimizer step: the oracle sees the same weights
        joint_parity, _ = joint_parity_vs_cpu(model, batch, cfg, parity_utts, dev)
    reducer = BucketedGradReducer(model)
    opt = FlatRAdam(model.named_parameters(), lr=1e-6, weight_decay=1e-6, reducer=reducer)   # tiny lr: the loss stays put

    last = {}

    def step():
        reducer.prepare()
        loss, losses_, _ = model.training_step(batch, global_step=10)
        last["losses"] = losses_
        loss.backward()
        reducer.finish()
        opt.clip_grad_norm(1.0)
        opt.step()
        return loss
    for _ in range(3):
        lv = step()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    ev[0].record()
    for i in range(steps):
        lv = step()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ms = float(np.median([ev[i].elapsed_time(ev[i + 1]) for i in range(steps)]))
    # host synchronisations of one step: torch's sync debug mode warns on every blocking device->host read
    n_sync = None
    try:
        torch.cuda.set_sync_debug_mode("warn")
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            step()
        n_sync = sum(1 for x in w if "synchroniz" in str(x.message).lower())
    finally:
        torch.cuda.set_sync_debug_mode("default")
    torch.cuda.synchronize()
    # where the step's time goes: HIP events at the section boundaries of one more step (forward sections by wrapping the
    # step's own methods; the backward is cut where the gradient of `context` -- the last thing the decoder's backward
    # produces -- is handed to the attention / text-encoder part)
    marks = {}

    def ev(name):
        marks[name] = torch.cuda.Event(enable_timing=True)
        marks[name].record()

    class _Mark(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x.view_as(x)

        @staticmethod
        def backward(ctx, g):
            ev("bwd_context_grad")
            return g

    def wrap(obj, name, before, after):
        fn = getattr(obj, name)

        def inner(*a, **k):
            ev(before)
            out = fn(*a, **k)
            ev(after)
            return out
        setattr(obj, name, inner)
        return lambda: delattr(obj, name) if name in obj.__dict__ else None
    undo = [wrap(model, "encode_text", "enc0", "enc1"), wrap(model, "compute_attention", "att0", "att1")]
    h1 = model.decoder.register_forward_pre_hook(
        lambda m, a, k: (ev("dec0"), ((a[0], a[1], _Mark.apply(a[2])) + tuple(a[3:]), k))[1], with_kwargs=True)
    h2 = model.decoder.register_forward_hook(lambda m, a, o: ev("dec1"))
    ev("t0")
    reducer.prepare()
    loss, _, _ = model.training_step(batch, global_step=10)
    ev("fwd_end")
    loss.backward()
    reducer.finish()
    ev("bwd_end")
    opt.clip_grad_norm(1.0)
    opt.step()
    ev("opt_end")
    torch.cuda.synchronize()
    for u in undo:
        u()
    h1.remove()
    h2.remove()
    # MAS: the default path takes the correctly rounded device log, the reference numpy's float32 log on the host
    # (alignment.py:36).  How many of THIS batch's alignments does that choice change?  (bit-equal search on equal logs;
    # INTEGRATION.md "MAS: which log")
    from rad_mmm_amd.alignment import binarize_attention
    with torch.no_grad():
        _, _, outs = model.training_step(batch, global_step=10)
        soft, il, ol = outs["attn_soft"], batch["input_lengths"], batch["output_lengths"]
        prev = os.environ.get("RADMMM_MAS_LOG")
        os.environ["RADMMM_MAS_LOG"] = "device"
        hard_dev = binarize_attention(soft, il, ol)
        os.environ["RADMMM_MAS_LOG"] = "host"
        hard_host = binarize_attention(soft, il, ol)
        if prev is None:
            del os.environ["RADMMM_MAS_LOG"]
        else:
            os.environ["RADMMM_MAS_LOG"] = prev
        mas_diff = int((hard_dev != hard_host).flatten(1).any(1).sum())
    dt_ = lambda a, b: round(marks[a].elapsed_time(marks[b]), 3)
    split = {"forward_text_encoder_ms": dt_("enc0", "enc1"), "forward_attention_mas_ms": dt_("att0", "att1"),
             "forward_decoder_ms": dt_("dec0", "dec1"), "forward_losses_nll_ctc_binarisation_ms": dt_("dec1", "fwd_end"),
             "backward_losses_and_decoder_ms": dt_("fwd_end", "bwd_context_grad"),
             "backward_attention_text_encoder_ms": dt_("bwd_context_grad", "bwd_end"),
             "clip_and_radam_ms": dt_("bwd_end", "opt_end"), "whole_instrumented_step_ms": dt_("t0", "opt_end")}
    jinfo = {}
    if joint:
        npred = sum(p.numel() for n, p in model.named_parameters() if "_predictor." in n)
        jinfo = {"joint": True, "predictors": sorted(JOINT_PREDICTORS), "predictor_parameters": npred,
                 "gradient_buckets": [b["key"] for b in reducer.buckets], "loss_terms": sorted(last.get("losses", {})),
                 "parity_vs_cpu": joint_parity}
    return {"what": ("BASELINE configs[3]: decoder + f0 / energy / voiced / duration predictors (ConvLSTMLinearDAP at the YAML dims, "
                     "dropout 0.5 on) in ONE step: " if joint else "") +
                    "TTSTrainingStep.training_step (text encoder + attention + on-device MAS + decoder + NLL/CTC/binarisation "
                    "losses) + backward + clip 1.0 + FlatRAdam; tts_lightning_modules.py:643-750", **jinfo,
            "batch": B, "frames": T, "text_tokens": t_txt, "steps": steps, "statistic": "median", "ms_per_step": ms,
            "value": B * T / (ms * 1e-3), "unit": "mel-frames/s", "loss": float(lv.detach()),
            "decoder_fwd_bwd_ms": decoder_only_ms, "ms_outside_decoder_fwd_bwd": ms - decoder_only_ms,
            "share_outside_decoder": (ms - decoder_only_ms) / ms, "host_syncs_per_step": n_sync, "split": split,
            "mas_alignments_changed_by_log_choice": {"differing": mas_diff, "of": B,
                                                     "what": "utterances whose hard alignment differs between the default device "
                                                             "log (correctly rounded fp32) and RADMMM_MAS_LOG=host (numpy's "
                                                             "float32 log, the reference's procedure, alignment.py:36)"}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--frames", type=int, default=800)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="radtts",
                    help="radtts = BASELINE configs[1] (headline); radmmm = configs[2] (shipped RADMMM decoder); "
                         "radmmm_splines = configs[4] architecture")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-throughput-mode", action="store_true",
                    help="skip the extra leg that times the 16-bit (single fp16 product) throughput mode")
    ap.add_argument("--optimizer", action="store_true",
                    help="also run the fused global-norm clip + RAdam update inside the timed step (NOT the BASELINE metric, "
                         "which is fwd+bwd only; reported with config.includes_optimizer = true)")
    ap.add_argument("--full-step", action="store_true",
                    help="extra leg (N = 1): the whole training step of the reference's TTSModel.training_step "
                         "(tts_lightning_modules.py:643-750: text encoder, alignment attention, on-device MAS, decoder, losses) "
                         "+ global-norm clip + RAdam on synthetic text/mel, reported under `full_step`")
    ap.add_argument("--rccl-channels", type=int, default=None,
                    help="N > 1: pin RCCL to this many channels (NCCL_MIN/MAX_NCHANNELS) and size the GEMM grids for the "
                         "remaining CUs; default 16 (rad_mmm_amd.ddp.reserve_collective_cus)")
    ap.add_argument("--step-only", action="store_true",
                    help="profiling aid (tools/prof_step.sh): run warm-up + timed steps, print ms per step and exit -- no roofline "
                         "leg, no extra legs, so that a kernel trace holds nothing but the steps")
    ap.add_argument("--kernel-only", action="store_true", help="time only the dominant kernel and exit")
    ap.add_argument("--dominant-only", action="store_true",
                    help="launch only the roofline leg's kernel (the PMC passes of tools/pmc_dominant.sh wrap this)")
    ap.add_argument("--joint-parity-utts", type=int, default=4,
                    help="--config joint: utterances of the bench batch the CPU oracle runs the whole joint step on")
    args = ap.parse_args()
    if args.config == "joint":
        args.full_step = True
    if args.dominant_only:
        import rad_mmm_amd  # noqa: F401
        torch.cuda.set_device(0)
        N, Tg = args.batch * (args.frames // 2), args.frames // 2
        npr = {"h3": 3, "f8x": 2}.get(os.environ.get("RADMMM_PRECISION", "f8x"), 2)      # the default product scheme
        hdur, hflop = time_dominant_kernel_h3(N, Tg, nprod=npr)
        print(json.dumps({"kernel": "rowgemm_h3 in_layer fwd", "nprod": npr, "M": N, "avg_launch_ms": hdur * 1e3,
                          "fp32_equiv_tflops": hflop / hdur / 1e12}))
        return
    if args.kernel_only:
        import rad_mmm_amd  # noqa: F401
        torch.cuda.set_device(0)
        N, Tg = args.batch * (args.frames // 2), args.frames // 2
        kdur, kflop = time_dominant_kernel(N, Tg)
        print(json.dumps({"kernel": "rowgemm_f32 in_layer fwd", "M": N, "avg_launch_ms": kdur * 1e3,
                          "tflops": kflop / kdur / 1e12, "tile_env": os.environ.get("RADMMM_ROWGEMM_TILE", "16")}))
        wdur, wflop = time_wgrad_kernel(N, Tg)
        print(json.dumps({"kernel": "wgrad_f32 in_layer", "R": N, "avg_launch_ms": wdur * 1e3,
                          "tflops": wflop / wdur / 1e12}))
        hdur, hflop = time_dominant_kernel_h3(N, Tg)
        print(json.dumps({"kernel": "rowgemm_h3 in_layer fwd", "M": N, "avg_launch_ms": hdur * 1e3,
                          "fp32_equiv_tflops": hflop / hdur / 1e12, "mfma_tflops": 3 * hflop / hdur / 1e12}))
        (w1, w2), wf = time_wgrad_h3(N, Tg)
        print(json.dumps({"kernel": "wgrad_h3 in_layer", "gemm_ms": w1 * 1e3, "gemm_plus_producers_ms": w2 * 1e3,
                          "fp32_equiv_tflops": wf / w1 / 1e12}))
        return

    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or os.environ.get("RADMMM_BENCH_SPAWN") == "1"):
        # bare `python bench.py --gpus N`: start the N ranks ourselves, one process per GPU, through torch's
        # launcher on the loopback address (the reference's Lightning `strategy: ddp`, `devices: auto` does the same:
        # configs/RADMMM_train_config.yaml:10,28).  RADMMM_BENCH_SPAWN=1 takes this path at N = 1 too (RCCL at world
        # size 1: the only multi-process check a 1-GPU box can run).
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
                   OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "8"), RADMMM_FORCE_DIST="1")
        sys.stdout.flush()
        os.execvpe(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                                    f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
                                    os.path.abspath(__file__), *sys.argv[1:]], env)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # test-only: RADMMM_BENCH_SHARE_GPU=1 puts every rank on device 0 and RADMMM_BENCH_BACKEND=gloo reduces through the host,
    # so that the control flow of an N > 1 run (who issues which collective when) can be exercised on a one-GPU box --
    # RCCL refuses two ranks on one device (tests/test_ddp_nccl.py)
    if os.environ.get("RADMMM_BENCH_SHARE_GPU") == "1":
        local_rank = 0
    backend = os.environ.get("RADMMM_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or os.environ.get("RADMMM_FORCE_DIST") == "1"   # the latter: RCCL smoke test on 1 GPU
    if world > 1:
        from rad_mmm_amd.ddp import reserve_collective_cus, RCCL_CUS   # before the library's first launch and before RCCL starts
        reserve_collective_cus(args.rccl_channels or RCCL_CUS)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))

    import rad_mmm_amd  # noqa: F401  (loads libradmmm_hip.so; no fallback)
    from rad_mmm_amd.common import SequenceLength
    from rad_mmm_amd.ddp import BucketedGradReducer
    from rad_mmm_amd.decoders import RADMMMFlow
    from rad_mmm_amd.loss import RADMMMLoss
    import radmmm_synth as O

    CFG = CONFIGS[args.config]
    cfg, sd = procedural_state(CFG)
    dec = RADMMMFlow(use_accent=True, **CFG)
    dec.load_state_dict(sd)
    dec = dec.to(dev).train()
    crit = RADMMMLoss(sigma=1.0, n_group_size=cfg.n_group_size)
    B, T = args.batch, args.frames
    batch = O.synthetic_batch(B, T, cfg, seed=1234 + rank, ragged=False)      # rank r: its own utterances
    gb = {k: torch.from_numpy(v).to(dev) for k, v in batch.items()}
    sl = SequenceLength(gb["lengths"])
    reducer = BucketedGradReducer(dec)
    reducer.profile = bool(use_dist)          # HIP events around the waits of finish(): exposed communication per step
    opt = None
    if args.optimizer:
        from rad_mmm_amd.optim import FlatRAdam
        opt = FlatRAdam(dec.named_parameters(), lr=1e-6, weight_decay=1e-6, reducer=reducer)   # tiny lr: loss stays put

    from rad_mmm_amd.ddp import reduce_loss_dict
    glob = {"handle": None}                   # the last step's coalesced loss-term reduce

    def step():
        reducer.prepare()
        out = dec(gb["mel"], gb["spk"], gb["context"], sl, gb["f0"], gb["energy"], gb["accent"])
        losses = crit(out, None, sl, 0)
        loss = losses["loss_mel"][0]
        if use_dist:
            # what the reference's loop does with `self.log(..., sync_dist=True)` per term and step
            # (tts_lightning_modules.py:746-749), as ONE asynchronous collective behind the forward pass; every rank, every step
            glob["handle"] = reduce_loss_dict(losses)
        loss.backward()
        reducer.finish()
        if opt is not None:
            opt.clip_grad_norm(1.0)
            opt.step()
        return loss

    for _ in range(args.warmup):
        loss = step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    # per-step HIP events on the compute stream (no synchronisation inside the timed region): the median step
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    comm_ev = []
    for i in range(args.steps):
        loss = step()
        marks[i + 1].record()
        comm_ev.append(reducer._prof_events)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    median_ms = step_ms[len(step_ms) // 2] if len(step_ms) % 2 else 0.5 * (step_ms[len(step_ms) // 2 - 1] + step_ms[len(step_ms) // 2])
    if use_dist:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt)
    ms_per_step = dt / args.steps * 1e3
    frames_per_s = world * B * T * args.steps / dt
    loss_val = float(loss.detach())
    # the mean over ranks of the last timed step's loss terms (rank 0 alone would print its own utterances' NLL)
    loss_global = ({k: float(v) for k, v in glob["handle"].wait().items()} if glob["handle"] is not None else None)
    # self-diagnosis of the data-parallel run: the world RCCL really spans (an all-reduce of ones), the time the compute
    # stream waited for all-reduces after backward (exposed communication; per bucket in issue order = last flow first),
    # the slowest rank's figure, and the knobs in force
    dist_info = {"backend": dist.get_backend() if use_dist else None, "process_group": bool(use_dist),
                 "gradient_buckets": len(reducer.buckets), "gradient_bytes_per_step": reducer.total_bytes,
                 "gemm_cu_budget": os.environ.get("RADMMM_GEMM_CUS"),
                 "nccl_env": {k: v for k, v in os.environ.items() if k.startswith(("NCCL_", "RCCL_"))}}
    if use_dist:
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)
        per_step = []
        per_bucket = None
        for ev in comm_ev:
            if ev:
                d = [ev[i].elapsed_time(ev[i + 1]) for i in range(len(ev) - 1)]
                per_step.append(sum(d))
                per_bucket = d if per_bucket is None else [a + b for a, b in zip(per_bucket, d)]
        exp_ms = float(np.median(per_step)) if per_step else None
        worst = torch.tensor([exp_ms or 0.0], device=dev, dtype=torch.float64)
        dist.all_reduce(worst, op=dist.ReduceOp.MAX)
        dist_info.update(rccl_world_size=dist.get_world_size(), rccl_allreduce_of_ones=float(ones),
                         reduce_op="avg" if reducer._avg else "sum + scale",
                         exposed_comm_ms_median_rank0=exp_ms, exposed_comm_ms_median_slowest_rank=float(worst),
                         exposed_comm_ms_per_bucket_mean_rank0=([x / len(per_step) for x in per_bucket] if per_bucket else None),
                         bucket_order=[reducer.buckets[i]["key"] for i in reducer._order],
                         bucket_mbytes=[round(reducer.buckets[i]["flat"].numel() * 4 / 1e6, 1) for i in reducer._order])
        # what the pinned channels deliver: every bucket all-reduced on its own (nothing else running), three times each
        # -- algorithm bandwidth = bytes / time, bus bandwidth = 2 (n - 1) / n of that (the ring's per-link load)
        bw = []
        scratch = {}
        for i in reducer._order:
            flat = reducer.buckets[i]["flat"]
            buf = scratch.setdefault(flat.numel(), torch.empty_like(flat))
            buf.copy_(flat)
            dist.all_reduce(buf)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                dist.all_reduce(buf)
            e1.record()
            torch.cuda.synchronize()
            t = torch.tensor([e0.elapsed_time(e1) / 3.0], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            gbs = flat.numel() * 4 / (float(t) * 1e-3) / 1e9
            bw.append({"bucket": reducer.buckets[i]["key"], "ms": round(float(t), 3), "algbw_gbs": round(gbs, 1),
                       "busbw_gbs": round(gbs * 2 * (world - 1) / max(world, 1), 1)})
        del scratch
        exposed_share = float(worst) / median_ms if median_ms else 0.0
        dist_info.update(allreduce_alone_per_bucket=bw, exposed_comm_share_of_step=exposed_share)
        if world > 1 and exposed_share > 0.05:
            note = (f"exposed gradient communication is {100 * exposed_share:.1f} % of the step (> 5 %): the "
                    f"{os.environ.get('NCCL_MAX_NCHANNELS', 'default')} RCCL channels this run is pinned to (rad_mmm_amd/ddp.py "
                    "reserve_collective_cus) under-deliver for these buckets, see allreduce_alone_per_bucket.  RCCL fixes its "
                    "channel count at the first communicator of a process, so the run cannot switch by itself: rerun with "
                    "--rccl-channels 32 (GEMM grids sized for 224 CUs) or with NCCL_MIN_NCHANNELS / NCCL_MAX_NCHANNELS unset "
                    "and RADMMM_GEMM_CUS=256 for RCCL's own choice")
            dist_info["rccl_channel_note"] = note
            if rank == 0:
                print("[bench] " + note, file=sys.stderr)

    if args.step_only:
        if rank == 0:
            print(json.dumps({"step_only": True, "ms_per_step": ms_per_step, "ms_per_step_median": median_ms, "steps": args.steps,
                              "warmup": args.warmup, "loss": loss_val}))
        return
    # the dominant kernel's launches INSIDE the training step, each bracketed by HIP events on its stream, over three more
    # steps -- on EVERY rank (a step issues the gradient all-reduces: rank 0 must not run one alone); rank 0 reports its own
    in_step = []
    if dec.gemm_precision in ("h3", "f8x"):
        from rad_mmm_amd import _lib as L
        Nrows = B * (T // cfg.n_group_size)
        L.LAUNCH_EVENTS.clear()
        L.LAUNCH_TIMER = lambda kw: (kw.get("taps") == 5 and kw.get("N") == 1024 and kw.get("K") == 1024 and kw.get("M") == Nrows
                                     and kw.get("Ch") is not None and kw.get("dact") is None and "act" in kw)
        for _ in range(3):
            step()
        L.LAUNCH_TIMER = None
        torch.cuda.synchronize()
        in_step = [a.elapsed_time(b) * 1e-3 for a, b in L.LAUNCH_EVENTS]
        L.LAUNCH_EVENTS.clear()

    def saturation_report():
        gs = getattr(dec, "_grad_scale", None)
        if gs is None:
            return None
        torch.cuda.synchronize()
        try:
            gs.check()                                # (consumes what is still in flight; under a process group it
        except FloatingPointError:                    #  all-reduces the flag words: every rank has to call it)
            pass
        return {"saturated_passes": gs.saturated_passes, "x8_saturated_passes": gs.x8_saturated_passes,
                "x8_adaptations": gs.x8_adaptations, "x8_grad_exp": gs.x8_grad_exp, "nonfinite_passes": gs.nonfinite_passes}
    sat_report = saturation_report() if use_dist else None      # (one GPU: taken at the end, after the side legs)
    if rank == 0:
        N = B * (T // cfg.n_group_size)
        h3 = dec.gemm_precision in ("h3", "f8x")
        f8x = dec.gemm_precision == "f8x"
        if h3:
            # split-f16 path: every fp32 product is three f16 MFMA products (Ah*Bh + Ah*Bl + Al*Bh) on the f16 matrix
            # cores.  `achieved` counts the ALGORITHMIC flops of the launch (2*M*N*K*taps) against the dense
            # f16 MFMA peak the kernel runs on; `executed_*` counts the 3x MFMA flops really issued (pipe utilisation).
            kdur_iso, kflop = time_dominant_kernel_h3(N, T // cfg.n_group_size, nprod=2 if f8x else 3)
            # ... and the same launches INSIDE the training step (the WN in_layer forward convs: same descriptor class),
            # each bracketed by HIP events on its stream, over three more steps.  Back to back the kernel runs into the
            # chip's power limit (MFMA at full tilt throttles the clock); in the step it alternates with memory-bound
            # kernels.  The in-step average is what rocprofv3's kernel trace of the step shows for this kernel
            # (profiles/r05_kernel_stats.json) and what `achieved` is priced on; the back-to-back figure stays beside it.
            kdur = float(np.mean(in_step)) if in_step else kdur_iso
            n_in_step = len(in_step)
            # executed MFMA work in f16-equivalent products: 3 f16 products, or 1 f16 + 2 FP8 products at twice the rate
            nprod, peak = (2.0 if f8x else 3.0), PEAK_F16_MFMA_TFLOPS
            kname = (("rowgemm_win_kernel<7,SPLIT> (rowgemm_win.hip: shared A window over the 5 taps; WN in_layer conv fwd, M=%d "
                      "N=1024 K=5x1024, hi.hi f16 MFMA + both cross terms in one block-scaled FP8 MFMA per 32-deep k step)" % N)
                     if f8x else
                     ("rowgemm_h3d_kernel<MB,3> (rowgemm_h3w.hip; WN in_layer conv fwd, M=%d N=1024 K=5x1024, 3 f16 MFMA "
                      "products per fp32 product)" % N))
            prec = ("f16 hi.hi product + FP8 (e4m3) cross terms, fp32 accumulate (this run's parity figures: `parity_vs_cpu`)"
                    if f8x else "split-f16 x3 MFMA products, fp32 accumulate (this run's parity figures: `parity_vs_cpu`)")
        else:
            kdur, kflop = time_dominant_kernel(N, T // cfg.n_group_size)
            kdur_iso, n_in_step = kdur, 0
            nprod, peak = 1.0, PEAK_FP32_MFMA_TFLOPS
            kname = "rowgemm_f32_kernel<0> (WN in_layer conv fwd, M=%d N=1024 K=5x1024)" % N
            prec = "fp32 MFMA"
        achieved = kflop / kdur / 1e12
        # HBM-side bytes per launch: PMC counters need rocprofv3 around the process, so they cannot be read in-run;
        # tools/pmc_dominant.sh measures this same launch (FETCH_SIZE / WRITE_SIZE in their own passes, the guide's
        # gfx950 corrections) and writes profiles/pmc_dominant.json, which is quoted here and labelled static
        traffic, traffic_src = None, None
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_dominant.json")) as f:
                pm = json.load(f)
            if h3 and pm.get("M") == N:
                traffic, traffic_src = pm["traffic_bytes_per_launch"], pm["source"]
        except Exception:
            pass
        fl = algorithmic_flops_per_frame(cfg)
        by = algorithmic_bytes_per_frame(cfg) + 3 * 4 * sum(p.numel() for p in dec.parameters()) / (B * T)
        res = {
            "metric": "mel-frames/sec training step (fwd+bwd)", "value": frames_per_s, "unit": "mel-frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "ms_per_step_median": median_ms, "ms_per_step_min": step_ms[0], "ms_per_step_max": step_ms[-1],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            # inputs, outputs, accumulation and the parity bar are fp32; the label names the multiplier arrays the products run on
            "dtype": ("f32 (f16 hi.hi product + fp8 e4m3 cross terms, fp32 accumulate)" if f8x else
                      "f32 (3 f16 products per fp32 product, fp32 accumulate)" if h3 else "f32 (fp32 MFMA)"),
            "data": "synthetic (procedural random-init weights, N(2.5,0.5) mel, fixed length)",
            "config": {"workload": ("RADTTS flow decoder (configs/RADTTS_model_config.yaml: 8 flows, WN 1024x4, "
                                    "D=1048) fwd+NLL+bwd") if args.config == "radtts" else
                                   ("RADMMM flow decoder (configs/RADMMM_model_config.yaml: 8 flows, WN 1024x4, D=1056) "
                                    "fwd+NLL+bwd") if args.config in ("radmmm", "joint") else
                                   ("RADMMM 16 kHz-dims flow decoder (configs/RADMMM_16khz_model_config.yaml + "
                                    "n_splines=2: 2 spline/FiLM + 6 affine/WN flows, D=1056) fwd+NLL+bwd"),
                       "batch_per_gpu": B, "n_mel": 80, "frames": T,
                       "global_batch": B * world, "parallelism": f"dp{world}", "precision": prec,
                       "includes_optimizer": bool(args.optimizer)},
            "loss_mel": loss_val,
            "loss_mel_global": (loss_global or {}).get("loss_mel"),
            "loss_terms_global": loss_global,
            "distributed": dist_info,
            "roofline": {"bound": "mfma", "kernel": kname,
                         "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                         "frac_algorithmic": achieved / peak, "frac_executed": nprod * achieved / peak,
                         "avg_launch_ms": kdur * 1e3, "avg_launch_measured": (f"HIP events around the {n_in_step} launches of this "
                                                                              "descriptor class in three training steps" if n_in_step
                                                                              else "HIP events around 20 back-to-back launches"),
                         "avg_launch_ms_back_to_back": kdur_iso * 1e3,
                         "flop_per_launch": kflop, "executed_mfma_flop_per_launch": nprod * kflop,
                         "executed_mfma_tflops": nprod * achieved,
                         # fp32 operands and result once = A 52 MB + W 21 MB + C 52 MB at M = 12 800
                         "algorithmic_bytes_per_launch": N * 1024 * 4 * 2 + 5 * 1024 * 1024 * 4,
                         # what this kernel's formats move at best: split A (hi + 8-bit cross array, 4 B/element), split
                         # weights (4 B) and the output's split pair (4 B; since round 5 no fp32 copy beside it under the
                         # FP8-cross scheme: 8 B with one), each once
                         "own_format_bytes_per_launch": (N * 1024 * (8 if f8x else 12) + 5 * 1024 * 1024 * 4) if h3 else None,
                         "traffic": traffic, "traffic_static": traffic is not None, "traffic_source": traffic_src,
                         # FETCH_SIZE counts L2 -> fabric requests, Infinity-Cache hits included (MI355X_MICROARCH.md, HBM): with 8
                         # XCDs = 8 private L2s an operand is fetched once per XCD that uses it.  The tile sequence gives an XCD
                         # r x c = 29 tiles (r row tiles of 224 rows x 4 KB, c column tiles of 5.2 MB of weights); r A + c B is
                         # minimal at c = 2 (the shipped order: r = 14.5): 13.3 + 10.5 MB per XCD = 190 MB of reads per launch
                         # for 73 MB of operands -- the floor of this counter for any 8-L2 mapping, not re-reads from HBM
                         "traffic_floor_8_private_l2": (190.4e6 + N * 1024 * (4.0 if f8x else 8.0)) if (h3 and N == 12800) else None,
                         # what a pure v_mfma_f32_32x32x16_f16 loop sustains on THIS data distribution (uniform random
                         # operands throttle the clock to ~1.55 GHz; zeros reach 2230): profiles/r01_mfma_dep.txt
                         "peak_measured_random_operands": 1620.0 if h3 else None,
                         "frac_executed_of_measured_peak": (nprod * achieved / 1620.0) if h3 else None},
            "step_flops": {"algorithmic_tflop_per_step": fl * B * T / 1e12,
                           "achieved_tflops_per_gpu": fl * B * T / (ms_per_step * 1e-3) / 1e12,
                           "frac_of_fp32_mfma_peak": fl * B * T / (ms_per_step * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS},
            "wn_stack": {"bound": "hbm (derived, north_star)", "algorithmic_gb_per_step": by * B * T / 1e9,
                         "achieved": by * B * T / (ms_per_step * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                         "frac": by * B * T / (ms_per_step * 1e-3) / 1e9 / PEAK_HBM_GBS},
        }
        def side_leg(mode, steps=5):
            """time `steps` steps with the decoder switched to another product scheme (same weights, same batch)"""
            prev = dec.gemm_precision
            dec.gemm_precision = mode
            os.environ["RADMMM_PRECISION"] = mode
            for _ in range(2):
                lv = step()
            torch.cuda.synchronize()
            # median of per-step HIP events, like the main leg (a five-step mean is at the mercy of one allocator hiccup)
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
            ev[0].record()
            for i in range(steps):
                lv = step()
                ev[i + 1].record()
            torch.cuda.synchronize()
            dts = float(np.median([ev[i].elapsed_time(ev[i + 1]) for i in range(steps)])) * 1e-3
            dec.gemm_precision = prev
            os.environ["RADMMM_PRECISION"] = prev
            return dts, float(lv.detach())

        if world == 1 and f8x and not args.no_throughput_mode:
            # the three-f16-product scheme (2e-6 instead of 4e-5 on z): same kernels, 3/2 of the MFMA work
            dt3, l3 = side_leg("h3")
            res["exact_split_mode"] = {"dtype": "split-f16 x3 MFMA products (RADMMM_PRECISION=h3), fp32 accumulate", "steps": 5,
                                       "ms_per_step": dt3 * 1e3, "statistic": "median", "value": B * T / dt3, "unit": "mel-frames/s", "loss_mel": l3,
                                       "loss_rel_diff_vs_default_mode": abs(l3 - loss_val) / abs(loss_val),
                                       "within_parity_bar": True}
        if world == 1 and h3 and not args.no_throughput_mode:
            # BASELINE's config label for this workload says "bf16": the same kernels with ONE fp16 MFMA product per
            # fp32 product (operands rounded to fp16, fp32 accumulate).  Reported beside, never as, `value`: this mode is
            # outside north_star's 1e-4 parity bar.
            dt16, l16 = side_leg("f16")
            res["throughput_mode"] = {"dtype": "f16 operands (single MFMA product), fp32 accumulate", "steps": 5,
                                      "ms_per_step": dt16 * 1e3, "statistic": "median", "value": B * T / dt16, "unit": "mel-frames/s",
                                      "loss_mel": l16, "loss_rel_diff_vs_parity_mode": abs(l16 - loss_val) / abs(loss_val),
                                      "within_parity_bar": False}
        hip_out = None
        if world == 1 and not args.no_cpu_baseline:
            # the decoder's outputs on the bench batch for `parity_vs_cpu`, taken BEFORE any leg that updates the weights
            # (the full-step leg and --optimizer run RAdam on them; the oracle runs on the initial state)
            if opt is not None:
                dec.load_state_dict(sd)
            with torch.no_grad():                     # (training mode, default scheme: the step's forward once more, outputs kept)
                hip_out = dec(gb["mel"], gb["spk"], gb["context"], sl, gb["f0"], gb["energy"], gb["accent"])
            hip_out = {k: (v.detach().clone() if torch.is_tensor(v) else
                           [t.detach().clone() if torch.is_tensor(t) else t for t in v] if isinstance(v, (list, tuple)) else v)
                       for k, v in hip_out.items()}
            torch.cuda.synchronize()
        if world == 1 and args.full_step:
            reducer.detach()                          # the step-wide reducer of that leg takes over the decoder's parameters
            res["full_step"] = full_step_leg(dec, cfg, CFG, gb, B, T, dev, median_ms, joint=args.config == "joint",
                                             parity_utts=0 if args.no_cpu_baseline else args.joint_parity_utts)
        # what the split producers reported over the whole run (ops.GradScale; published without host synchronisation)
        res["saturation"] = sat_report if sat_report is not None else saturation_report()
        res["roofline_hbm"] = hbm_rooflines(dec, cfg, B, T)
        res["roofline_mfma"] = mfma_rooflines(dec, cfg, B, T)
        # what the process-group path costs a step at world size 1 (an A/B of two processes on one box cannot
        # be taken from inside one run: quoted from the committed measurement, with what it was made of)
        res["process_group_overhead_ms"] = {
            "value": 0.5, "static": True, "source": "profiles/r06_pg_overhead.txt",
            "what": "RADMMM_BENCH_SPAWN=1 (RCCL at world size 1, bucketed all-reduce inside the step) minus the default run, same box, "
                    "alternating: 41.66 / 41.67 against 41.14 / 41.22 ms; was +1.8 ms (41.84 / 41.69 against 39.97 / 39.93) until the "
                    "one-rank ReduceOp.AVG -- RCCL's oneRankReduce<FuncPreMulSum>: 17 launches rewriting 876 MB beside the GEMMs, "
                    "1.31 ms of kernel time per step -- became an in-place SUM at world size 1 (rad_mmm_amd/ddp.py)"}
        if hip_out is not None:
            res["cpu_baseline"], res["parity_vs_cpu"] = cpu_baseline(cfg, sd, batch, hip_out)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # RCCL prints its version banner through C stdio: flush it first so the JSON is the last line
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
