"""Monotonic alignment search on the GPU (reference alignment.py:31-59 and
TTSModel.binarize_attention, tts_lightning_modules.py:270-284).

The reference copies the [B, T_mel, T_txt] attention to the host and runs a numba loop per
item; here the whole batch is one kernel launch and nothing leaves the device.  The DP uses the
same fp32 additions in the same order, so on identical log inputs the result is bit-exact
(tests/test_hip_aux.py checks that on all 32 items of a benchmark-size batch with numpy's log).
The log itself: the reference takes numpy's float32 log on the host, whose vectorised implementation
is only accurate to a few ulp and differs between CPUs; the batched path here takes the CORRECTLY
ROUNDED fp32 log inside the kernel (radmmm_mas_width1_prob) -- a device-independent definition.  A log
differing in the last bit can only flip an exact near-tie; the test counts how often that happens.

RADMMM_MAS_LOG=host (a supported switch, read per call) makes `binarize_attention` take the log exactly as the reference
does -- numpy's float32 log of the attention on the host -- and run the device search on it: bit-exact with the reference
on the same machine's numpy, at the price of one device -> host -> device round trip of the [B, T_mel, T_txt] map per
step.  INTEGRATION.md recommends the default (device log): training does not depend on which way an exact tie of the
alignment falls, and the default is reproducible across hosts.
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops


def binarize_attention(attn: torch.Tensor, in_lens: torch.Tensor, out_lens: torch.Tensor) -> torch.Tensor:
    """attn [B, 1, T_mel, T_txt] soft attention -> hard 0/1 attention of the same shape."""
    import os
    with torch.no_grad():
        B, _, T1, T2 = attn.shape
        il, ol = in_lens.to(torch.int32).contiguous(), out_lens.to(torch.int32).contiguous()
        if os.environ.get("RADMMM_MAS_LOG", "device") == "host":
            with np.errstate(divide="ignore"):
                logp = torch.from_numpy(np.log(attn[:, 0].detach().float().cpu().numpy())).to(attn.device)
            return ops.mas_width1_batch(logp, il, ol)[:, None].to(attn.dtype)
        hard = ops.mas_width1_batch(attn[:, 0], il, ol, prob=True)
        return hard[:, None]


def mas_width1(attn_map: np.ndarray, device: str = "cuda:0") -> np.ndarray:
    """numpy [T_mel, T_txt] in, numpy 0/1 out (the reference's signature).  The log is taken on the
    host with numpy exactly as the reference does."""
    with np.errstate(divide="ignore"):
        logp = np.log(attn_map.astype(np.float32))
    t = torch.from_numpy(logp)[None].to(device)
    T1, T2 = attn_map.shape
    hard = ops.mas_width1_batch(t, torch.tensor([T2], dtype=torch.int32, device=device),
                                torch.tensor([T1], dtype=torch.int32, device=device))
    return hard[0].cpu().numpy().astype(attn_map.dtype)
