#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/*.npz by RUNNING THE REFERENCE.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py [--ref /root/reference]

The reference is pure Python; it is imported unmodified from its checkout with
three tiny stand-ins for packages this image lacks and that do not take part in
the arithmetic being captured:
  * numba.jit          -> identity decorator (alignment.py then runs as plain
                          Python: same semantics, the JIT only compiles it)
  * librosa.util.pad_center / tiny, librosa.filters.mel (mel raises: the
    filterbank is NOT captured -> "parity unpinned")
Nothing of the reference's source is written into this repo: fixtures hold
inputs, weights (random/procedural, generated here) and the outputs the
reference computed for them.
"""
import argparse
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def install_stubs():
    numba = types.ModuleType("numba")
    numba.jit = lambda *a, **k: (lambda f: f)
    sys.modules["numba"] = numba

    librosa = types.ModuleType("librosa")
    util = types.ModuleType("librosa.util")
    filters = types.ModuleType("librosa.filters")

    def pad_center(data, size, axis=-1, **kw):
        n = data.shape[axis]
        lpad = int((size - n) // 2)
        lengths = [(0, 0)] * data.ndim
        lengths[axis] = (lpad, int(size - n - lpad))
        return np.pad(data, lengths)

    def tiny(x):
        return np.finfo(np.asarray(x).dtype).tiny

    def mel(*a, **k):
        raise RuntimeError("librosa absent: mel filterbank is not captured")

    util.pad_center, util.tiny = pad_center, tiny
    filters.mel = mel
    librosa.util, librosa.filters = util, filters
    sys.modules["librosa"] = librosa
    sys.modules["librosa.util"] = util
    sys.modules["librosa.filters"] = filters


def t2n(d):
    return {k: (v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)) for k, v in d.items()}


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrs)
    print(f"wrote {name}: {os.path.getsize(path) / 1024:.1f} KiB")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    install_stubs()
    sys.path[:0] = [args.ref, os.path.join(args.ref, "vocoders")]
    cwd = os.getcwd()
    os.chdir("/tmp")
    import torch
    import torch.nn.functional as F
    torch.set_num_threads(8)
    import common
    import splines
    import loss as ref_loss
    import alignment
    import audio_processing
    import decoders
    from oracle import radmmm_oracle as O
    os.chdir(cwd)
    only = set(args.only.split(",")) if args.only else None
    want = lambda n: only is None or n in only

    def rand_mask_inputs(B, C, Cctx, T, lens, seed):
        g = torch.Generator().manual_seed(seed)
        z = torch.randn(B, C, T, generator=g)
        ctx = torch.randn(B, Cctx, T, generator=g)
        lens = torch.tensor(lens)
        return z, ctx, common.SequenceLength(lens)

    # ------------------------------------------------------------------ WN + affine coupling
    if want("affine"):
        torch.manual_seed(11)
        layer = common.AffineTransformationLayer(
            8, 12, 4, affine_model="wavenet", scaling_fn="tanh", affine_activation="softplus",
            n_channels=16, use_partial_padding=True)
        wn = layer.affine_param_predictor
        with torch.no_grad():
            wn.end.weight.normal_(0, 0.2)
            wn.end.bias.normal_(0, 0.1)
            for n, p in layer.named_parameters():
                if n.endswith("weight_g"):
                    p.mul_(1.0 + 0.2 * torch.randn_like(p))
        z, ctx, sl = rand_mask_inputs(3, 8, 12, 37, [37, 30, 21], 5)
        z.requires_grad_(True)
        ctx.requires_grad_(True)
        zo, log_s = layer(z, ctx, seq_lens=sl)
        mask = sl.mask[:, None].float()
        scalar = 0.5 * ((zo * mask) ** 2).sum() - (log_s * mask).sum()
        scalar.backward()
        arrs = {"in.z": z, "in.ctx": ctx, "in.lens": sl.lengths, "out.z": zo, "out.log_s": log_s,
                "out.scalar": scalar, "grad.z": z.grad, "grad.ctx": ctx.grad}
        for n, p in layer.state_dict().items():
            arrs["sd." + n] = p
        for n, p in layer.named_parameters():
            arrs["gradp." + n] = p.grad
        # also the bare WN output.  (seq_lens=None is not captured: PartialConv1d
        # caches mask_ratio by input shape, partialconv1d.py:65, so an unmasked call
        # after a masked one silently reuses the stale ratio; the decoder always
        # passes seq_lens, decoders.py:190-191.)
        with torch.no_grad():
            arrs["out.wn"] = wn((z[:, :4], ctx), seq_lens=sl)
            arrs["out.z_inverse"] = layer(zo, ctx, inverse=True, seq_lens=sl)
        save("affine_tiny.npz", **t2n(arrs))

    # ------------------------------------------------------------------ invertible 1x1
    if want("inv1x1"):
        torch.manual_seed(12)
        lus = common.Invertible1x1ConvLUS(8)
        with torch.no_grad():
            lus.upper_diag.mul_(1.0 + 0.3 * torch.randn(8))
        g = torch.Generator().manual_seed(3)
        z = torch.randn(3, 8, 19, generator=g, requires_grad=True)
        zo, ld = lus(z)
        cot = torch.randn(3, 8, 19, generator=g)
        ((zo * cot).sum() + 3.0 * ld).backward()
        arrs = {"lus.in.z": z, "lus.cot": cot, "lus.out.z": zo, "lus.out.logdet": ld,
                "lus.grad.z": z.grad}
        for n, p in lus.state_dict().items():
            arrs["lus.sd." + n] = p
        for n, p in lus.named_parameters():
            arrs["lus.gradp." + n] = p.grad
        with torch.no_grad():
            arrs["lus.out.z_inverse"] = lus(zo, inverse=True)

        wh = common.DataInitializedInvertible1x1Conv(8)
        wh.train()
        lens = torch.tensor([40, 33, 25])
        zz = 2.5 + torch.randn(3, 8, 40, generator=g) * torch.linspace(0.5, 1.5, 8)[None, :, None]
        zz = zz.clone().requires_grad_(True)
        zo, ld = wh(zz, lens=common.SequenceLength(lens))
        cot = torch.randn(3, 8, 40, generator=g)
        ((zo * cot).sum() + 2.0 * ld).backward()
        arrs.update({"wh.in.z": zz, "wh.in.lens": lens, "wh.cot": cot, "wh.out.z": zo,
                     "wh.out.logdet": ld, "wh.grad.z": zz.grad})
        for n, p in wh.state_dict().items():
            arrs["wh.sd." + n] = p
        for n, p in wh.named_parameters():
            arrs["wh.gradp." + n] = p.grad
        save("inv1x1_tiny.npz", **t2n(arrs))

    # ------------------------------------------------------------------ splines
    if want("spline"):
        g = torch.Generator().manual_seed(21)
        N, k, K = 50, 4, 32
        x = torch.rand(N, k, generator=g) * 8 - 4          # some outside [-3,3)
        xn = ((x + 3) / 6).requires_grad_(True)
        wt = torch.randn(N, k, K, generator=g, requires_grad=True)
        vt = torch.randn(N, k, K + 1, generator=g, requires_grad=True)
        y, lj = splines.unbounded_piecewise_quadratic_transform(xn, wt, vt)
        cot = torch.randn(N, k, generator=g)
        ((y * cot).sum() + (lj * torch.flip(cot, [0])).sum()).backward()
        arrs = {"pq.in.x": xn, "pq.in.w": wt, "pq.in.v": vt, "pq.cot": cot, "pq.out.y": y,
                "pq.out.logj": lj, "pq.grad.x": xn.grad, "pq.grad.w": wt.grad, "pq.grad.v": vt.grad}
        with torch.no_grad():
            xi, _ = splines.unbounded_piecewise_quadratic_transform(y.detach(), wt, vt, inverse=True)
            arrs["pq.out.x_inverse"] = xi

        torch.manual_seed(22)
        layer = common.SplineTransformationLayer(
            8, 12, 2, scaling_fn="tanh", top=3, bottom=-3, left=-3, right=3, n_bins=32,
            use_quadratic=True, use_bn=True)
        layer.train()
        pp = layer.param_predictor
        # shrink the FiLM hidden width is not possible through the ctor (512 fixed):
        # keep it, but store only what the oracle needs (state is ~2.7M floats -> too
        # big); instead re-draw weights procedurally from names/shapes.
        shapes = {n: tuple(p.shape) for n, p in layer.state_dict().items()}
        proc = O.procedural_decoder_state(shapes, end_scale=0.05)
        layer.load_state_dict({n: torch.from_numpy(np.asarray(v)) for n, v in proc.items()})
        z, ctx, sl = rand_mask_inputs(3, 8, 12, 29, [29, 22, 17], 23)
        z = (z * 1.5).requires_grad_(True)
        ctx.requires_grad_(True)
        zo, log_s = layer(z, ctx, seq_lens=sl)
        mask = sl.mask[:, None].float()
        scalar = 0.5 * ((zo * mask) ** 2).sum() - (log_s * mask).sum()
        scalar.backward()
        arrs.update({"sp.in.z": z, "sp.in.ctx": ctx, "sp.in.lens": sl.lengths, "sp.out.z": zo,
                     "sp.out.log_s": log_s, "sp.out.scalar": scalar, "sp.grad.z": z.grad,
                     "sp.grad.ctx": ctx.grad})
        for n, s in shapes.items():
            arrs["sp.shape." + n] = np.asarray(s, dtype=np.int64)
        for n, p in layer.named_parameters():
            if p.numel() <= 2048:
                arrs["sp.gradp." + n] = p.grad
            else:
                arrs["sp.gradnorm." + n] = p.grad.norm()
        save("spline_tiny.npz", **t2n(arrs))

    # ------------------------------------------------------------------ flow loss
    if want("loss"):
        g = torch.Generator().manual_seed(31)
        lens = torch.tensor([40, 31, 22])
        gsz = 2
        z = torch.randn(3, 12, 20, generator=g)
        log_s = [torch.randn(3, 6, 20, generator=g), torch.randn(3, 5, 20, generator=g),
                 torch.randn(3, 1, 20, generator=g)]
        ldw = [torch.tensor(0.3), torch.tensor(-1.2), torch.tensor(0.05)]
        crit = ref_loss.RADMMMLoss(sigma=0.9, n_group_size=gsz)
        n_el = torch.div(lens.sum(), gsz, rounding_mode="floor")
        mask = common.get_mask_from_lengths(torch.div(lens, gsz, rounding_mode="floor"))[:, None].float()
        lm, lp = ref_loss.compute_flow_loss(z, [t.clone() for t in ldw], log_s, n_el, 12, mask, 0.9)
        save("flow_loss.npz", **t2n({"z": z, "log_s0": log_s[0], "log_s1": log_s[1], "log_s2": log_s[2],
                                     "ldw": torch.stack(ldw), "lens": lens, "loss_mel": lm,
                                     "loss_prior": lp}))

    # ------------------------------------------------------------------ attention / MAS / CTC
    if want("attention"):
        torch.manual_seed(41)
        att = common.ConvAttention(8, 16, 8)
        g = torch.Generator().manual_seed(42)
        B, T1, T2 = 3, 30, 11
        out_lens = torch.tensor([30, 24, 19])
        in_lens = torch.tensor([11, 9, 7])
        q = torch.randn(B, 8, T1, generator=g, requires_grad=True)
        k = torch.randn(B, 16, T2, generator=g, requires_grad=True)
        prior = torch.rand(B, T1, T2, generator=g) + 0.05
        prior = prior / prior.sum(2, keepdim=True)
        kmask = ~common.get_mask_from_lengths(in_lens)[..., None]
        attn, lp = att(q, k, out_lens, kmask, key_lens=in_lens, attn_prior=prior)
        ctc = ref_loss.AttentionCTCLoss()(lp, in_lens, out_lens)
        hard = np.zeros(attn.shape, dtype=np.float32)
        a_np = attn.detach().numpy()
        for b in range(B):
            hard[b, 0, :out_lens[b], :in_lens[b]] = alignment.mas_width1(
                a_np[b, 0, :out_lens[b], :in_lens[b]])
        hard_t = torch.from_numpy(hard)
        binl = ref_loss.AttentionBinarizationLoss()(hard_t, attn)
        (ctc + 0.7 * binl).backward()
        arrs = {"in.q": q, "in.k": k, "in.prior": prior, "in.out_lens": out_lens, "in.in_lens": in_lens,
                "out.attn": attn, "out.logprob": lp, "out.ctc": ctc, "out.hard": hard_t,
                "out.bin": binl, "grad.q": q.grad, "grad.k": k.grad}
        for n, p in att.state_dict().items():
            arrs["sd." + n] = p
        for n, p in att.named_parameters():
            arrs["gradp." + n] = p.grad
        # extra MAS cases incl. near-ties, sharp and flat maps, 1-column map
        r = np.random.Generator(np.random.PCG64(7))
        for ci, (a, b, sharp) in enumerate([(50, 17, 1.0), (64, 64, 4.0), (33, 1, 1.0), (120, 40, 0.2),
                                            (9, 9, 0.0)]):
            logits = r.standard_normal((a, b)).astype(np.float32) * sharp
            m = np.exp(logits - logits.max(1, keepdims=True))
            m = (m / m.sum(1, keepdims=True)).astype(np.float32)
            if sharp == 0.0:
                m[:] = np.float32(1.0 / b)      # all ties
            arrs[f"mas.{ci}.in"] = m
            arrs[f"mas.{ci}.out"] = alignment.mas_width1(m.copy())
        save("attention_tiny.npz", **t2n(arrs))

    # ------------------------------------------------------------------ STFT
    if want("stft"):
        st = audio_processing.STFT(64, 16, 64)
        r = np.random.Generator(np.random.PCG64(51))
        audio = np.clip(r.standard_normal((2, 400)) * 0.3, -1, 1).astype(np.float32)
        mag, _ = st.transform(torch.from_numpy(audio))
        st2 = audio_processing.STFT(1024, 256, 1024)
        audio2 = np.clip(r.standard_normal((1, 4096)) * 0.3, -1, 1).astype(np.float32)
        mag2, _ = st2.transform(torch.from_numpy(audio2))
        save("stft_tiny.npz", audio=audio, mag=mag.numpy(), audio2=audio2,
             mag2=mag2.numpy().astype(np.float32))

    # ------------------------------------------------------------------ full decoder, procedural weights
    def run_decoder(tag, cfg_kwargs, B, T, ragged, keep_grads):
        cfg = O.DecoderConfig(**cfg_kwargs)
        ref_kwargs = dict(cfg_kwargs)
        ref_kwargs.setdefault("use_accent", True)
        dec = decoders.RADMMMFlow(**ref_kwargs)
        dec.train()
        shapes = {n: tuple(p.shape) for n, p in dec.state_dict().items()}
        mine = O.decoder_state_shapes(cfg)
        assert shapes == mine, (set(shapes) ^ set(mine),
                                [(k, shapes[k], mine[k]) for k in shapes if k in mine and shapes[k] != mine[k]])
        proc = O.procedural_decoder_state(shapes)
        dec.load_state_dict({n: torch.from_numpy(np.asarray(v)) for n, v in proc.items()})
        batch = O.synthetic_batch(B, T, cfg, seed=1234, ragged=ragged)
        tb = {k: torch.from_numpy(v) for k, v in batch.items()}
        sl = common.SequenceLength(tb["lengths"])
        mel = tb["mel"].clone().requires_grad_(True)
        ctx = tb["context"].clone().requires_grad_(True)
        out = dec(mel, tb["spk"], ctx, sl, tb["f0"], tb["energy"], tb["accent"])
        # capture BEFORE the loss: compute_flow_loss accumulates in place into
        # log_det_W_list[0] (loss.py:93-101), mutating the decoder's output list
        log_det_W = torch.stack([t.detach().clone() for t in out["log_det_W_list"]])
        crit = ref_loss.RADMMMLoss(sigma=1.0, n_group_size=cfg.n_group_size)
        n_el = torch.div(sl.lengths.sum(), cfg.n_group_size, rounding_mode="floor")
        mask = common.get_mask_from_lengths(torch.div(sl.lengths, cfg.n_group_size,
                                                      rounding_mode="floor"))[:, None].float()
        lm, lp = ref_loss.compute_flow_loss(out["z_mel"], out["log_det_W_list"], out["log_s_list"],
                                            n_el, out["z_mel"].size(1), mask, 1.0)
        lm.backward()
        arrs = {"B": B, "T": T, "ragged": ragged, "lengths": batch["lengths"],
                "z_mel": out["z_mel"].detach().numpy().astype(np.float32),
                "log_det_W": log_det_W, "loss_mel": lm, "loss_prior": lp,
                "grad.mel": mel.grad, "ctx_w_spkvec.sum": out["context_w_spkvec"].sum(),
                "ctx_w_spkvec.slice": out["context_w_spkvec"][:, :8, :16]}
        for i, ls in enumerate(out["log_s_list"]):
            arrs[f"log_s.{i}.masked_sum"] = (ls * mask).sum()
            arrs[f"log_s.{i}.slice"] = ls[:, :4, :32]
        arrs["grad.context.norm"] = ctx.grad.norm()
        arrs["grad.context.slice"] = ctx.grad[:, :8, :32]
        for n, p in dec.named_parameters():
            arrs["gradnorm." + n] = p.grad.norm()
            if p.numel() <= 4096 or n in keep_grads:
                arrs["gradp." + n] = p.grad
            elif p.dim() == 3:
                arrs["gradslice." + n] = p.grad[:4, :8].contiguous()
        for k, v in cfg_kwargs.items():
            arrs["cfg." + k] = np.asarray(v)
        save(f"decoder_{tag}.npz", **t2n(arrs))

    radtts = dict(n_speaker_dim=16, use_accent_emb_for_decoder=True, n_accent_dim=8, n_text_dim=512,
                  n_f0_dims=1, n_energy_avg_dims=1, n_mel_channels=80, n_early_size=2, n_early_every=2,
                  n_group_size=2, scaling_fn="tanh", affine_activation="softplus",
                  use_partial_padding=True, n_conv_layers_per_step=4)
    if want("decoder_cfg1"):
        # BASELINE config 1: 2 flow steps, batch 2, T<=256 (ragged), CPU reference path
        run_decoder("cfg1", dict(radtts, n_flows=2), 2, 256, True, set())
    if want("decoder_cfg2s"):
        # config-2 architecture (8 flows, D=1048) at a CPU-sized batch
        run_decoder("cfg2_small", dict(radtts, n_flows=8), 2, 96, True, set())
    if want("decoder_cfg5s"):
        # config-5 architecture: RADMMM/16 kHz dims (n_text_dim 520, accent not in decoder), 2 spline steps
        mm = dict(radtts, n_text_dim=520, use_accent_emb_for_decoder=False, n_flows=4, n_splines=2,
                  use_bn=True)
        run_decoder("cfg5_small", mm, 2, 64, True, set())
    if want("decoder_cfg3s"):
        # BASELINE configs[2]: the shipped RADMMM decoder (configs/RADMMM_model_config.yaml:16-39): 8 affine flows,
        # n_text_dim 520, accent embedding NOT fed to the decoder -> D = 1056, at a CPU-sized batch
        run_decoder("cfg3_small", dict(radtts, n_text_dim=520, use_accent_emb_for_decoder=False, n_flows=8),
                    2, 112, True, set())

    # ------------------------------------------------------------------ text Encoder (f1)
    if want("encoder"):
        for tag, norm in (("plain", None), ("spectral", "spectral")):
            torch.manual_seed(31)
            enc = common.Encoder(encoder_n_convolutions=3, encoder_embedding_dim=32, encoder_kernel_size=5,
                                 lstm_norm_fn=norm)
            with torch.no_grad():
                for n, p in enc.named_parameters():
                    if n.endswith(".1.weight"):
                        p.copy_(1.0 + 0.3 * torch.randn_like(p))
                    elif n.endswith(".1.bias"):
                        p.copy_(0.2 * torch.randn_like(p))
            enc.eval()                                       # dropout off; spectral norm uses the stored u, v
            g = torch.Generator().manual_seed(8)
            lens = torch.tensor([11, 7, 2, 9])
            x = torch.randn(4, 32, 11, generator=g)
            for b in range(4):
                x[b, :, int(lens[b]):] = 0
            x.requires_grad_(True)
            out = enc(x, lens)
            gw = torch.randn(out.shape, generator=g)
            (out * gw).sum().backward()
            arrs = {"x": x, "lens": lens, "out": out, "gw": gw, "grad.x": x.grad}
            for n, t in enc.state_dict().items():
                arrs["sd." + n] = t
            for n, p in enc.named_parameters():
                arrs["gradp." + n] = p.grad
            save(f"encoder_{tag}.npz", **t2n(arrs))

    # ------------------------------------------------------------------ training_step glue (a17)
    if want("tts_step"):
        # tts_lightning_modules needs pytorch_lightning (absent), so the step is composed here from the reference's
        # own components in the order of TTSModel.training_step (tts_lightning_modules.py:643-750)
        n_text = 32
        kw = dict(radtts, n_text_dim=n_text, n_flows=2)
        dec = decoders.RADMMMFlow(use_accent=True, **kw)
        enc = common.Encoder(3, n_text, 5, lstm_norm_fn=None)
        mods = torch.nn.ModuleDict(dict(
            text_embeddings=torch.nn.Embedding(40, n_text), text_encoder=enc,
            speaker_embeddings=torch.nn.Embedding(3, 16), accent_embeddings=torch.nn.Embedding(2, 8),
            attention=common.ConvAttention(80, n_text), decoder=dec))
        shapes = {n: tuple(p.shape) for n, p in mods.state_dict().items()}
        proc = O.procedural_decoder_state(shapes)
        mods.load_state_dict({n: torch.from_numpy(np.asarray(v)) for n, v in proc.items()})
        mods.train()
        for m in mods.modules():                      # deterministic step: no dropout
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
        enc_drop = F.dropout
        g = torch.Generator().manual_seed(21)
        B, T, L = 2, 48, 9
        out_lens = torch.tensor([48, 36])
        in_lens = torch.tensor([9, 6])
        batch = {"mel": torch.randn(B, 80, T, generator=g) * 1.2 - 5.0, "speaker_ids": torch.tensor([0, 2]),
                 "accent_ids": torch.tensor([1, 0]), "text": torch.randint(0, 40, (B, L), generator=g),
                 "input_lengths": in_lens, "output_lengths": out_lens,
                 "attn_prior": torch.rand(B, T, L, generator=g) * 0.9 + 0.05, "f0": torch.rand(B, T, generator=g) * 6,
                 "energy_avg": torch.rand(B, T, generator=g)}
        for b in range(B):
            batch["mel"][b, :, int(out_lens[b]):] = 0
            batch["f0"][b, int(out_lens[b]):] = 0
            batch["energy_avg"][b, int(out_lens[b]):] = 0
            batch["text"][b, int(in_lens[b]):] = 0
        crit = ref_loss.RADMMMLoss(sigma=1.0, kl_loss_start_iter=5)
        crit.n_group_size = dec.n_group_size
        arrs = {("batch." + k): v for k, v in batch.items()}
        real_dropout = F.dropout
        F.dropout = lambda x, p=0.5, training=True, inplace=False: x      # Encoder calls F.dropout(x, 0.5, self.training)
        try:
            for tag, step, binarize in (("soft", 0, False), ("hard", 10, True)):
                mods.zero_grad()
                il, ol = common.SequenceLength(in_lens), common.SequenceLength(out_lens)
                mel = (batch["mel"] + 5) / 2
                spk = mods["speaker_embeddings"](batch["speaker_ids"])
                acc = mods["accent_embeddings"](batch["accent_ids"])
                emb = mods["text_embeddings"](batch["text"]).transpose(1, 2)
                txt_enc = mods["text_encoder"](emb, il.lengths).transpose(1, 2)
                attn_mask = common.get_mask_from_lengths(il.lengths)[..., None] == 0
                attn_soft, attn_logprob = mods["attention"](mel, emb, ol.lengths, attn_mask, key_lens=il.lengths,
                                                            attn_prior=batch["attn_prior"])
                if binarize:
                    hard = torch.zeros_like(attn_soft)
                    a_np = attn_soft.data.cpu().numpy()
                    for i in range(B):
                        hard[i, 0, :out_lens[i], :in_lens[i]] = torch.tensor(
                            alignment.mas_width1(a_np[i, 0, :out_lens[i], :in_lens[i]]))
                    attn = hard
                else:
                    attn = attn_soft
                context = torch.bmm(txt_enc, attn.squeeze(1).transpose(1, 2))
                outputs = mods["decoder"](mel, spk, context, ol, f0=batch["f0"], energy_avg=batch["energy_avg"],
                                          accent_vecs=acc)
                outputs.update(attn=attn, attn_soft=attn_soft, attn_logprob=attn_logprob)
                ld = crit(outputs, il, ol, step)
                loss = None
                for k, (v, w) in ld.items():
                    loss = v * w if loss is None else loss + v * w
                    arrs[f"{tag}.{k}"] = v if torch.is_tensor(v) else torch.tensor(float(v))
                loss.backward()
                arrs[f"{tag}.loss"] = loss
                arrs[f"{tag}.attn"] = attn
                arrs[f"{tag}.context"] = context
                for n, p in mods.named_parameters():
                    if p.grad is not None and (n.startswith(("text_", "speaker_", "accent_", "attention")) or "context_lstm" in n):
                        arrs[f"{tag}.gradnorm.{n}"] = p.grad.norm()
        finally:
            F.dropout = real_dropout
        for k, v in kw.items():
            arrs["cfg." + k] = np.asarray(v)
        save("tts_step.npz", **t2n(arrs))

    # ------------------------------------------------------------------ attribute predictor (f2)
    if want("dap"):
        import attribute_predictors as ref_ap
        torch.manual_seed(17)
        dap = ref_ap.ConvLSTMLinearDAP(n_speaker_dim=16, in_dim=32, out_dim=1, reduction_factor=4, n_backbone_layers=2,
                                       n_hidden=16, kernel_size=3, p_dropout=0.25, target_scale=2.0, target_offset=0.5,
                                       log_target=True, lstm_type="bilstm", use_speaker_embedding=True)
        dap.eval()
        g = torch.Generator().manual_seed(5)
        lens = torch.tensor([13, 9, 2, 13])
        sl = common.SequenceLength(lens)
        txt = torch.randn(4, 32, 13, generator=g)
        for b in range(4):
            txt[b, :, int(lens[b]):] = 0
        txt.requires_grad_(True)
        spk = torch.randn(4, 16, generator=g)
        target = torch.rand(4, 1, 13, generator=g) * 3
        out = dap(target, txt, spk, sl)
        crit = ref_loss.AttributeRegressionLoss(prefix="f0_", weight=1.0)
        loss = crit(out, None, sl, 0)["f0_loss"][0]
        loss.backward()
        arrs = {"txt": txt, "spk": spk, "lens": lens, "target": target, "x_hat": out["x_hat"], "x": out["x"], "loss": loss,
                "grad.txt": txt.grad}
        for n, t in dap.state_dict().items():
            arrs["sd." + n] = t
        for n, p in dap.named_parameters():
            arrs["gradp." + n] = p.grad
        save("dap_tiny.npz", **t2n(arrs))

    # ------------------------------------------------------------------ decoder.infer (inverse flows, f4)
    if want("infer"):
        # decoders.py:221 allocates the noise with torch.cuda.FloatTensor; on this GPU-less box the CPU
        # type stands in (same normal_() stream from torch's CPU generator, seeded below)
        torch.cuda.FloatTensor = torch.FloatTensor
        for tag, nfl, T_txt in (("cfg1", 2, 9), ("cfg2_small", 8, 7), ("cfg5_small", 4, 8)):
            cfg_kwargs = dict(radtts, n_flows=nfl)
            if tag == "cfg5_small":      # RADMMM / 16 kHz dims, 2 spline + 2 affine flows (eval: running BN stats)
                cfg_kwargs = dict(radtts, n_text_dim=520, use_accent_emb_for_decoder=False, n_flows=4, n_splines=2,
                                  use_bn=True)
            cfg = O.DecoderConfig(**cfg_kwargs)
            dec = decoders.RADMMMFlow(use_accent=True, **cfg_kwargs)
            shapes = {n: tuple(p.shape) for n, p in dec.state_dict().items()}
            # end_scale 0.002: the coupling scales stay in (0.7, 1.3); with the forward fixtures' 0.02 a random
            # (untrained) model has scales near 0 and the inverse of noise overflows
            proc = O.procedural_decoder_state(shapes, end_scale=0.002)
            dec.load_state_dict({n: torch.from_numpy(np.asarray(v)) for n, v in proc.items()})
            dec.eval()
            g = torch.Generator().manual_seed(99 + nfl)
            B = 2
            dur = torch.randint(1, 6, (B, T_txt), generator=g)
            dur[1, -2:] = 0                                    # shorter second utterance
            for b in range(B):                                 # even frame counts (group size 2)
                if int(dur[b].sum()) % 2:
                    dur[b, 0] += 1
            out_lens = dur.sum(1)
            Tmax = int(out_lens.max())
            txt_enc = torch.randn(B, cfg.n_text_dim, T_txt, generator=g)
            spk = torch.randn(B, cfg.n_speaker_dim, generator=g)
            acc = torch.randn(B, cfg.n_accent_dim, generator=g)
            f0 = torch.rand(B, Tmax, generator=g) * 6
            en = torch.rand(B, Tmax, generator=g)
            for b in range(B):
                f0[b, int(out_lens[b]):] = 0
                en[b, int(out_lens[b]):] = 0
            torch.manual_seed(4242)
            with torch.no_grad():
                out = dec.infer(spk, txt_enc, 0.8, dur=dur, f0=f0, energy_avg=en, out_lens=out_lens, accent_vecs=acc)
            arrs = {"dur": dur, "out_lens": out_lens, "txt_enc": txt_enc, "spk": spk, "accent": acc, "f0": f0,
                    "energy": en, "sigma": 0.8, "seed": 4242, "end_scale": 0.002, "mel": out["mel"]}
            for k, v in cfg_kwargs.items():
                arrs["cfg." + k] = np.asarray(v)
            save(f"infer_{tag}.npz", **t2n(arrs))

    # ------------------------------------------------------------------ RAdam + global-norm clip (f3)
    if want("radam"):
        import radam as ref_radam
        g = torch.Generator().manual_seed(77)
        shapes = [(7, 5), (33,), (4, 3, 5), (1,)]
        params = [torch.nn.Parameter(torch.randn(*sh, generator=g)) for sh in shapes]
        p0 = [p.detach().clone() for p in params]
        opt = ref_radam.RAdam(params, lr=1e-3, weight_decay=1e-6)
        n_steps = 9                                    # N_sma crosses 5 between steps 5 and 6
        grads = [[torch.randn(*sh, generator=g) * (3.0 if k % 3 == 0 else 0.05) for sh in shapes] for k in range(n_steps)]
        arrs = {}
        for i, t in enumerate(p0):
            arrs[f"p0.{i}"] = t
        for k in range(n_steps):
            for p, gr in zip(params, grads[k]):
                p.grad = gr.clone()
            total = torch.nn.utils.clip_grad_norm_(params, 1.0)     # Lightning gradient_clip_val 1.0, algorithm norm
            opt.step()
            arrs[f"norm.{k}"] = total
            for i, (p, gr) in enumerate(zip(params, grads[k])):
                arrs[f"g.{k}.{i}"] = gr
                arrs[f"p.{k}.{i}"] = p.detach().clone()
        save("radam_tiny.npz", **t2n(arrs))

    # ------------------------------------------------------------------ data path (f4): attention prior + energy
    if want("prior"):
        # data.py pulls in packages that take no part in this arithmetic (lmdb cache, pyin f0 extraction,
        # praat augmentation, text cleaners); scipy.stats.betabinom / scipy.ndimage.zoom, which do, are the real ones
        for name in ("lmdb", "parselmouth", "parselmouth.praat", "wave_transforms", "tts_text_processing",
                     "tts_text_processing.text_processing"):
            if name not in sys.modules:
                sys.modules[name] = types.ModuleType(name)
        sys.modules["librosa"].pyin = None
        sys.modules["parselmouth.praat"].call = None
        sys.modules["wave_transforms"].WaveAugmentations = None
        sys.modules["tts_text_processing.text_processing"].TextProcessing = None
        import data as ref_data
        arrs = {}
        pairs = [(7, 23), (1, 1), (2, 9), (39, 160), (61, 333), (150, 799), (12, 1000)]     # (n_tokens, n_frames)
        interp = ref_data.BetaBinomialInterpolator()
        for i, (p, m) in enumerate(pairs):
            arrs[f"interp.{i}.pm"] = np.array([p, m])
            arrs[f"interp.{i}.out"] = interp(p, m)
        for i, (p, m, sc) in enumerate([(5, 9, 0.05), (20, 100, 0.05), (33, 47, 1.0), (1, 4, 0.05)]):
            arrs[f"bank.{i}.pms"] = np.array([p, m, sc])
            arrs[f"bank.{i}.out"] = ref_data.beta_binomial_prior_distribution(p, m, sc)
        g = torch.Generator().manual_seed(5)
        mel = torch.randn(80, 37, generator=g) * 2.0 - 5.0
        arrs["energy.mel"] = mel
        for scaled in (True, False):
            ns = types.SimpleNamespace(use_scaled_energy=scaled)
            ns.energy_avg_normalize = lambda x, ns=ns: ref_data.AudioDataset.energy_avg_normalize(ns, x)
            arrs[f"energy.out.{int(scaled)}"] = ref_data.AudioDataset.get_energy_average(ns, mel)
        save("prior.npz", **t2n(arrs))

    # ------------------------------------------------------------------ embedding regularisers + BCE loss (caller, a17)
    if want("regloss"):
        import loss as ref_loss
        import common as ref_common
        gen = torch.Generator().manual_seed(91)
        spk = torch.nn.Embedding(7, 16)
        acc = torch.nn.Embedding(3, 8)
        with torch.no_grad():
            spk.weight.copy_(torch.randn(7, 16, generator=gen) * 0.7 + 0.1)
            acc.weight.copy_(torch.randn(3, 8, generator=gen) * 1.3)
        sid = torch.tensor([0, 3, 3, 6, 1])
        aid = torch.tensor([2, 0, 1, 1, 2])
        arrs = {"spk": spk.weight, "acc": acc.weight, "sid": sid, "aid": aid}
        vc = ref_loss.VarianceCovarianceEmbeddingRegLoss("speaker", 0.3, 0.7, gamma=1.0)(spk)
        arrs["vc.variance"], arrs["vc.covariance"] = vc["loss_speaker_variance"][0], vc["loss_speaker_covariance"][0]
        vt = ref_loss.VarianceCovarianceEmbeddingRegLoss("accent", 1.0, 1.0, gamma=2.0)(acc)    # (modules only: a tensor raises)
        arrs["vt.variance"], arrs["vt.covariance"] = vt["loss_accent_variance"][0], vt["loss_accent_covariance"][0]
        cc = ref_loss.AttributeMinCrossCovarianceRegLoss("speaker", "accent", 1.0)
        arrs["cc.tables"] = cc(spk(sid), acc(aid), spk, acc)["loss_speaker-accent_cross_covariance"][0]
        arrs["cc.batch"] = cc(spk(sid), acc(aid), None, None)["loss_speaker-accent_cross_covariance"][0]
        x = (torch.rand(3, 1, 11, generator=gen) > 0.5).float()
        x_hat = torch.randn(3, 1, 11, generator=gen) * 2
        lens = ref_common.SequenceLength(torch.tensor([11, 4, 7]))
        arrs["bce.x"], arrs["bce.x_hat"], arrs["bce.lens"] = x, x_hat, lens.lengths
        arrs["bce.loss"] = ref_loss.AttributeBCELoss("vpred_", 1.0)({"x": x, "x_hat": x_hat}, None, lens, 0)["vpred_loss"][0]
        save("regloss.npz", **t2n(arrs))


if __name__ == "__main__":
    main()
