"""Helper of tests/test_ddp_nccl.py::test_gradient_exchange_values_at_world_size_two (run under torch.distributed.run with
two ranks, BOTH on cuda:0, gradients exchanged by gloo through the host -- RCCL refuses two ranks on one device and a GPU
box has one GPU).

What it proves (VERDICT r4 "missing 1"): the reference averages real gradients over its ranks (Lightning `strategy: ddp`,
configs/RADMMM_train_config.yaml:28).  Here every rank runs the REAL decoder (WN width 1024, the wide split kernels: 4800
rows) on ITS OWN utterances through `BucketedGradReducer` with direct gradient sinks and the early bucket start
(ops.notify_grads_final -> ddp._grads_final: a flow step's upper-layer bucket is all-reduced half a flow step before its
autograd node returns).  At world size 1 an in-place all-reduce is the identity, so a bucket launched before its last write
still ends up right; with two ranks it does not.  Checked per parameter, two steps in a row on different batches:

    .grad after finish()  ==  0.5 * (single-rank gradient of rank 0 + single-rank gradient of rank 1)

where the single-rank gradients come from a twin decoder's plain backward (no reducer) and are averaged by a synchronous
all-reduce of CPU tensors after that backward has completed.  Then (`spline` case) one spline flow with masked batch-norm
synchronised over the ranks (maskedbatchnorm1d.py:88-95, TTSModel.toggle_syncbnorm) against the single-process run on the
concatenated batch.  `--negative` is the control that shows the check has teeth: the lower-half buckets are announced final
together with the upper halves, i.e. BEFORE their gradients are written, and the same comparison must fail."""
import argparse
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

KW = dict(n_speaker_dim=16, use_accent_emb_for_decoder=True, n_accent_dim=8, n_text_dim=512, n_f0_dims=1,
          n_energy_avg_dims=1, n_mel_channels=80, n_early_size=2, n_early_every=2, n_group_size=2,
          scaling_fn="tanh", affine_activation="softplus", use_partial_padding=True, n_conv_layers_per_step=4, n_flows=2)


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--negative", action="store_true")
    args = ap.parse_args()
    import radmmm_synth as S
    from rad_mmm_amd import ops
    from rad_mmm_amd.common import SequenceLength
    from rad_mmm_amd.ddp import BucketedGradReducer
    from rad_mmm_amd.decoders import RADMMMFlow
    from rad_mmm_amd.loss import RADMMMLoss
    from rad_mmm_amd.spline_layers import toggle_syncbnorm
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    assert world == 2
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    crit = RADMMMLoss(n_group_size=2)

    def build(kw):
        cfg = S.DecoderConfig(**kw)
        sd = {k: torch.from_numpy(np.asarray(v)) for k, v in S.procedural_decoder_state(S.decoder_state_shapes(cfg)).items()}
        dec = RADMMMFlow(use_accent=True, **kw)
        dec.load_state_dict(sd)
        return cfg, dec.to(dev).train()

    def batch_of(cfg, B, T, seed, ragged):
        return {k: torch.from_numpy(v).to(dev) for k, v in S.synthetic_batch(B, T, cfg, seed=seed, ragged=ragged).items()}

    def step(dec, b):
        sl = SequenceLength(b["lengths"])
        out = dec(b["mel"], b["spk"], b["context"], sl, b["f0"], b["energy"], b["accent"])
        loss = crit(out, None, sl, 0)["loss_mel"][0]
        loss.backward()
        return loss.detach()

    # ------------------------------------------------------------------ affine flows: exact mean of the single-rank passes
    cfg, twin = build(KW)
    _, dec = build(KW)
    red = BucketedGradReducer(dec)
    assert red.active and red.world == 2 and any(b["key"].endswith(".hi") for b in red.buckets)
    n_direct = sum(1 for bk in red.buckets for p in bk["params"] if red._direct[id(p)])
    assert n_direct > 40
    early_marks = []
    inner = red._grads_final

    def counted(ptrs):
        if args.negative and red.active:
            # control: declare the SAME flow step's lower-half bucket final as well -- its gradients (layers 1, 0, the start
            # conv, the channel mix) have not been written yet, so its all-reduce starts on stale data
            # (flow 1 only: every parameter of its lower half writes through its sink -- flow 0's whitening conv goes through
            #  stock autograd, which the reducer's own guard would refuse to pair with an early start)
            key = red._by_param[id(red._by_ptr[ptrs[0]])]["key"]
            for bk in red.buckets:
                if key.startswith("flows.1.") and bk["key"] == key[:-3] + ".lo" and not bk["ready"]:
                    for p in bk["params"]:
                        red._early.add(id(p))
                    bk["pending"], bk["ready"] = 0, True
        inner(ptrs)
        early_marks.append(len(red._early))
    red._grads_final = counted
    worst, worst_name = 0.0, ""
    for it in range(2):
        b = batch_of(cfg, 12, 800, seed=100 + 10 * it + rank, ragged=True)           # rank r: its own utterances
        for p in twin.parameters():
            p.grad = None
        step(twin, b)
        torch.cuda.synchronize()
        names = [n for n, _ in twin.named_parameters()]
        local = {n: p.grad.detach().float().cpu() for n, p in twin.named_parameters()}
        flat = torch.cat([local[n].reshape(-1) for n in names])
        dist.all_reduce(flat)                                                        # synchronous, CPU tensors, after backward
        flat *= 0.5
        red.prepare()
        step(dec, b)
        red.finish()
        torch.cuda.synchronize()
        off = 0
        for n, p in dec.named_parameters():
            want = flat[off: off + p.numel()].view_as(p)
            off += p.numel()
            e = rel(p.grad.detach().cpu(), want)
            if e > worst:
                worst, worst_name = e, n
        # the reducer's gradient must also DIFFER from this rank's own (the ranks hold different utterances)
        own = max(rel(p.grad.detach().cpu(), local[n]) for n, p in dec.named_parameters() if p.numel() > 1000)
        assert own > 1e-3, f"ranks' gradients do not differ ({own:.1e}): the check would be vacuous"
    assert early_marks and max(early_marks) > 0, "the early bucket start was never taken"
    if args.negative:
        print(f"DDP_WORLD2_NEGATIVE rank={rank} worst={worst:.3e} ({worst_name})", flush=True)
        assert worst > 1e-6, "control failed: premature all-reduce was not detected"
    else:
        assert worst <= 1e-6, (worst, worst_name)
        print(f"DDP_WORLD2_AFFINE_OK rank={rank} worst={worst:.3e} early_marked={max(early_marks)} direct={n_direct}", flush=True)
    red.detach()
    del dec, twin, red
    torch.cuda.empty_cache()

    if not args.negative:
        # -------------------------------------------------------------- one spline flow, synchronised masked batch-norm
        kw = dict(KW, n_splines=1, use_bn=True)
        cfg, ref = build(kw)
        _, dec = build(kw)
        toggle_syncbnorm(dec, True)
        red = BucketedGradReducer(dec)
        parts = [batch_of(cfg, 12, 800, seed=300 + r, ragged=False) for r in range(world)]
        whole = {k: torch.cat([pt[k] for pt in parts], 0) for k in parts[0]}
        step(ref, whole)                                                             # single process, concatenated batch
        red.prepare()
        step(dec, parts[rank])
        red.finish()
        torch.cuda.synchronize()
        worst_l2, worst_name, worst_sp, worst_sp_name = 0.0, "", 0.0, ""
        gref = dict(ref.named_parameters())
        for n, p in dec.named_parameters():
            # (scale and bias of the conv that feeds a batch-norm have an analytically ZERO gradient -- the normalisation
            #  removes both: what is left is rounding residue, not comparable in relative terms)
            if n.endswith(("hidden_conv.conv.weight_g", "hidden_conv.conv.bias")):
                continue
            a, w = p.grad.detach().double(), gref[n].grad.detach().double()
            e = float((a - w).norm() / w.norm().clamp_min(1e-30))
            # the spline flow's own parameters and everything upstream of it (the context LSTM) see the log-Jacobian's KINK at
            # the knots: synchronised statistics (partial sums all-reduced) and the concatenated run's differ in the last bit,
            # an element within an ulp of a bin edge then takes the neighbouring bin -- continuous in z, O(1) in that element's
            # parameter gradient, ~1e-3 of the tensor (profiles/r03_spline_bin_edge_ties.txt).  The affine flow behind
            # it only sees z: held to 2e-4.  A wrong exchange (stale data, a missing 1/2) is O(1) on every tensor.
            if n.startswith(("flows.0.", "context_lstm.")):
                if e > worst_sp:
                    worst_sp, worst_sp_name = e, n
            elif e > worst_l2:
                worst_l2, worst_name = e, n
        assert worst_l2 <= 2e-4, (worst_l2, worst_name)
        assert worst_sp <= 5e-3, (worst_sp, worst_sp_name)
        print(f"DDP_WORLD2_SPLINE_SYNCBN_OK rank={rank} affine_flow_worst_l2={worst_l2:.3e} ({worst_name}) "
              f"spline_flow_and_upstream_worst_l2={worst_sp:.3e} ({worst_sp_name})", flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
