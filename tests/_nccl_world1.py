"""Helper of tests/test_ddp_nccl.py (run as a script on a GPU box): the real decoder's backward through
BucketedGradReducer with direct gradient sinks, once without a process group and once with RCCL (backend "nccl") at
world size 1 -- gradients must be bit-identical (a world-size-1 all-reduce is the identity, the mean divides by 1)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from oracle import radmmm_oracle as O
    from rad_mmm_amd.common import SequenceLength
    from rad_mmm_amd.ddp import BucketedGradReducer
    from rad_mmm_amd.decoders import RADMMMFlow
    from rad_mmm_amd.loss import RADMMMLoss
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    kw = dict(n_speaker_dim=16, use_accent_emb_for_decoder=True, n_accent_dim=8, n_text_dim=512, n_f0_dims=1,
              n_energy_avg_dims=1, n_mel_channels=80, n_early_size=2, n_early_every=2, n_group_size=2,
              scaling_fn="tanh", affine_activation="softplus", use_partial_padding=True,
              n_conv_layers_per_step=4, n_flows=3)
    cfg = O.DecoderConfig(**kw)
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in O.procedural_decoder_state(O.decoder_state_shapes(cfg)).items()}
    b = {k: torch.from_numpy(v).to(dev) for k, v in O.synthetic_batch(4, 128, cfg, 11, ragged=True).items()}
    crit = RADMMMLoss(n_group_size=2)

    def run(steps):
        dec = RADMMMFlow(use_accent=True, **kw)
        dec.load_state_dict(sd)
        dec = dec.to(dev).train()
        red = BucketedGradReducer(dec)
        for _ in range(steps):
            red.prepare()
            sl = SequenceLength(b["lengths"])
            out = dec(b["mel"], b["spk"], b["context"], sl, b["f0"], b["energy"], b["accent"])
            crit(out, None, sl, 0)["loss_mel"][0].backward()
            red.finish()
        torch.cuda.synchronize()
        direct = sum(1 for bk in red.buckets for p in bk["params"] if red._direct[id(p)])
        return {n: p.grad.detach().cpu().clone() for n, p in dec.named_parameters()}, red, direct

    g0, red0, _ = run(2)
    assert not red0.active
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    g1, red1, direct = run(2)
    assert red1.active and direct > 0
    assert all(bk["handle"] is not None for bk in red1.buckets)         # every bucket went through RCCL
    bad = [n for n in g0 if not torch.equal(g0[n], g1[n])]
    assert not bad, bad[:5]
    dist.barrier()
    dist.destroy_process_group()
    print(f"NCCL_WORLD1_OK buckets={len(red1.buckets)} direct_params={direct} params={len(g0)}")


if __name__ == "__main__":
    main()
