"""Flow NLL + attention losses (reference: loss.py).

`RADMMMLoss` / `RADTTSLoss` keep the reference's constructor keywords and the
`forward(model_output, in_lens, out_lens, global_step) -> {name: (value, weight)}` contract
(loss.py:518-537, 192-211).  The masked reductions of the NLL run as HIP kernels; the CTC
term stays on torch.nn.CTCLoss.
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import nn
import torch.nn.functional as F

from . import ops
from .common import SequenceLength
from ._trace import traced


def compute_flow_loss(z, log_det_W_list, log_s_list, n_elements, n_dims, lens32, sigma=1.0):
    """loss.py:85-110 with mask = [t < lens32[b]] applied inside the reduction kernels
    (no log 2*pi term, as in the reference).  Unlike the reference this does not mutate
    log_det_W_list[0] in place."""
    log_s_total = None
    for log_s in log_s_list:
        s = ops.masked_sum(log_s, lens32)
        log_s_total = s if log_s_total is None else log_s_total + s
    log_det_W_total = 0.0
    if len(log_det_W_list):
        log_det_W_total = torch.stack(list(log_det_W_list)).sum() * n_elements
    prior_NLL = ops.masked_sumsq(z, lens32) / (2 * sigma * sigma)
    loss = prior_NLL - log_s_total - log_det_W_total
    denom = n_elements * n_dims
    return loss / denom, prior_NLL / denom


class AttentionCTCLoss(nn.Module):
    """loss.py:112-141: blank column (log-prob -1) in front, per-utterance log_softmax over ITS text positions, torch's
    CTC with targets 1 .. len, zero_infinity, mean over the batch of (loss / target length).  The reference loops over
    the utterances with two host reads each (`int(in_lens[bid])`): 64 synchronisations per step at B = 32.  Here the batch
    goes through ONE F.ctc_loss call: text positions beyond an utterance's length are taken out of its softmax with a
    -1e4 logit (exp underflows to exactly 0 in fp32, so every remaining value is the per-utterance softmax's; a literal
    -inf would turn torch's CTC gradient, exp(lp) - exp(alpha*beta - lp), into NaN at those positions), frames beyond its
    mel length are ignored through input_lengths.  Same values as the loop (tests/golden/tts_step.npz).  On the GPU the
    CTC itself is radmmm_ctc_monotonic (csrc/ctc.hip): the targets are always 1 .. len, so the 2 len + 1 states are one
    THREAD each, the alpha and the beta chain of an utterance run side by side in two workgroups (a double-buffered LDS
    row, one barrier per frame) and a second, elementwise launch writes torch's gradient formula from the stored alpha /
    beta ([2][B][T][2 C - 1] floats of scratch: ~65 MB at B = 32, T = 800, 150 tokens) -- two launches, no host
    synchronisation; a one-wave-per-utterance version with the states in registers was built first and dropped at 5.6 ms (F.ctc_loss copies its length arguments between host and device: six blocking copies per step even
    with host lengths).  The -1e4 mask assumes every real logit lies well above -1e4 + 88 (the attention's
    log-probabilities are >= ~-1e3)."""

    def __init__(self, blank_logprob=-1):
        super().__init__()
        self.blank_logprob = blank_logprob

    def forward(self, attn_logprob, in_lens, out_lens, in_lens_host=None, out_lens_host=None):
        padded = F.pad(attn_logprob, (1, 0), value=self.blank_logprob)[:, 0]          # [B, T_mel, 1 + T_txt]
        B, _, C = padded.shape
        cls = torch.arange(C, device=padded.device)
        lp = padded.masked_fill(cls[None, None, :] > in_lens[:, None, None], -1e4)
        lp = torch.log_softmax(lp, -1)                                               # [B, T_mel, C]
        if lp.is_cuda and C - 1 <= 511:
            # the targets are every text position once, in order: radmmm_ctc_monotonic (one launch for value and gradient,
            # torch's formula; no host synchronisation -- F.ctc_loss makes six per step -- and 0.3 instead of 2.3 ms)
            loss = ops.CTCMonotonicFn.apply(lp, in_lens.to(torch.int32).contiguous(), out_lens.to(torch.int32).contiguous())
        else:
            targets = cls[1:][None].expand(B, -1)                                    # 1 .. T_txt; the first len count
            loss = F.ctc_loss(lp.transpose(0, 1), targets, out_lens if out_lens_host is None else out_lens_host,
                              in_lens if in_lens_host is None else in_lens_host, blank=0, reduction="none", zero_infinity=True)
        return (loss / in_lens.clamp_min(1).to(loss.dtype)).sum() / B


class AttentionBinarizationLoss(nn.Module):
    """loss.py:143-151: binary cross entropy (target 1, mean) of the soft attention at the positions the hard alignment
    selects.  The reference gathers them with a boolean mask (`soft[hard == 1]`: a device -> host read of the count); the
    same mean as a masked sum -- -sum(hard * max(log soft, -100)) / sum(hard), torch's BCE clamps its log at -100 -- needs
    none (hard is exactly 0 / 1)."""

    def forward(self, hard_attention, soft_attention):
        on = hard_attention == 1
        # positions the alignment does not select never reach the log (their value is replaced by 1 first): an exact 0 there
        # (every padded text column of the masked softmax, an underflowed probability) would otherwise give log's backward
        # 0 / 0 = NaN although its weight `sel` is 0 -- the reference's boolean gather cannot see those elements at all
        # (torch's own BCE on the substituted tensor: its value clamps log at -100 and its gradient is
        #  (x - 1) / max(x (1 - x), 1e-12) -- finite at a SELECTED probability of exactly 0 too, as in the reference's call)
        # (F.binary_cross_entropy is on autocast's banned list -- "unsafe to autocast", whatever the input dtype -- so under
        #  Lightning's bf16-mixed the call would raise from the first step past kl_loss_start_iter on: the term runs with
        #  autocast off on an fp32 copy, like every kernel of this package; tests/test_tts_step.py covers it)
        with torch.autocast(device_type=soft_attention.device.type, enabled=False):
            soft = soft_attention.float()
            sel = on.to(soft.dtype)
            ones = torch.ones_like(soft)
            bce = F.binary_cross_entropy(torch.where(on, soft, ones), ones, reduction="none")
            return (sel * bce).sum() / sel.sum()


class AttentionLoss(nn.Module):
    """loss.py:153-179."""

    def __init__(self, CTC_blank_logprob=-1, kl_loss_start_iter=5000, binarization_loss_weight=1.0,
                 ctc_loss_weight=0.1):
        super().__init__()
        self.attn_ctc_loss = AttentionCTCLoss(blank_logprob=CTC_blank_logprob)
        self.attn_bin_loss = AttentionBinarizationLoss()
        self.kl_loss_start_iter = kl_loss_start_iter
        self.binarization_loss_weight = binarization_loss_weight
        self.ctc_loss_weight = ctc_loss_weight

    def forward(self, attn, attn_soft, attn_logprob, global_step, in_lens, out_lens, in_lens_host=None, out_lens_host=None):
        loss_dict = {"loss_ctc": (self.attn_ctc_loss(attn_logprob, in_lens, out_lens, in_lens_host, out_lens_host),
                                  self.ctc_loss_weight)}
        if global_step > self.kl_loss_start_iter:
            loss_dict["binarization_loss"] = (self.attn_bin_loss(attn, attn_soft), self.binarization_loss_weight)
        else:
            loss_dict["binarization_loss"] = (0.0, self.binarization_loss_weight)
        return loss_dict


class RADTTSLoss(nn.Module):
    """loss.py:182-211."""

    def __init__(self, sigma=1.0, n_group_size=1, CTC_blank_logprob=-1, kl_loss_start_iter=5000,
                 binarization_loss_weight=1.0, ctc_loss_weight=0.1):
        super().__init__()
        self.sigma = sigma
        self.n_group_size = n_group_size
        self.attn_loss = AttentionLoss(CTC_blank_logprob, kl_loss_start_iter, binarization_loss_weight,
                                       ctc_loss_weight)

    @traced("loss")
    def forward(self, model_output, in_lens: Optional[SequenceLength], out_lens: SequenceLength, global_step):
        loss_dict = {}
        if len(model_output["z_mel"]):
            g = self.n_group_size
            n_elements = torch.div(out_lens.lengths.sum(), g, rounding_mode="floor")
            lens32 = torch.div(out_lens.lengths, g, rounding_mode="floor").to(torch.int32)
            n_dims = model_output["z_mel"].size(1)
            loss_mel, loss_prior_mel = compute_flow_loss(
                model_output["z_mel"], model_output["log_det_W_list"], model_output["log_s_list"], n_elements,
                n_dims, lens32, self.sigma)
            loss_dict["loss_mel"] = (loss_mel, 1.0)
            loss_dict["loss_prior_mel"] = (loss_prior_mel, 0.0)
        # The reference indexes model_output['attn*'] unconditionally (loss.py:206-208); the
        # decoder-only harness (bench, parity tests) has no aligner, so the term is optional here.
        if "attn_logprob" in model_output:
            loss_dict.update(self.attn_loss(model_output["attn"], model_output["attn_soft"],
                                            model_output["attn_logprob"], global_step, in_lens.lengths,
                                            out_lens.lengths, getattr(in_lens, "lengths_host", None),
                                            getattr(out_lens, "lengths_host", None)))
        return loss_dict


class RADMMMLoss(RADTTSLoss):
    """loss.py:500-537: same arithmetic as RADTTSLoss; the extra constructor keywords are stored and, as in the
    reference, never read by forward (the embedding regularisers are separate modules called by the training step:
    VarianceCovarianceEmbeddingRegLoss / AttributeMinCrossCovarianceRegLoss below, tts_step.py)."""

    def __init__(self, sigma=1.0, n_group_size=1, CTC_blank_logprob=-1, kl_loss_start_iter=5000,
                 binarization_loss_weight=1.0, ctc_loss_weight=0.1, use_spk_embed_reg=False,
                 use_accent_embed_reg=False, reg_loss_config=None, use_spk_accent_cross_covariance=False,
                 cross_reg_loss_config=None):
        super().__init__(sigma, n_group_size, CTC_blank_logprob, kl_loss_start_iter, binarization_loss_weight,
                         ctc_loss_weight)
        self.use_spk_embed_reg = bool(use_spk_embed_reg)
        self.use_accent_embed_reg = bool(use_accent_embed_reg)
        self.use_spk_accent_cross_covariance = bool(use_spk_accent_cross_covariance)
        self.reg_loss_config = reg_loss_config
        self.cross_reg_loss_config = cross_reg_loss_config


class AttributeBCELoss(nn.Module):
    """loss.py:213-230 (the voiced predictor's loss): masked binary cross-entropy on logits, summed and divided by
    the number of valid positions -> {prefix + 'loss': (value, weight)}.  A few KB of work: stock torch ops."""

    def __init__(self, prefix: Optional[str] = None, weight=1.0):
        super().__init__()
        self.prefix, self.weight = prefix, weight

    def forward(self, model_output, in_lens, out_lens, global_step, mask=None):
        target, prediction = model_output["x"], model_output["x_hat"]
        if mask is None:
            mask = out_lens.mask.unsqueeze(1)
        assert mask.dim() == target.dim()
        mask = mask.bool().expand_as(prediction)                      # masked sum instead of the reference's boolean gather: no
        zero = torch.zeros_like(prediction)                            # device -> host read of the count
        bce = F.binary_cross_entropy_with_logits(torch.where(mask, prediction, zero), torch.where(mask, target, zero),
                                                 reduction="none")
        loss = torch.where(mask, bce, zero).sum() / mask.sum()
        return {self.prefix + "loss": (loss, self.weight)}


def _table(emb):
    """an nn.Embedding (its weight) or a plain [n, d] tensor"""
    return emb.weight if isinstance(emb, nn.Module) else emb


class VarianceCovarianceEmbeddingRegLoss(nn.Module):
    """loss.py:314-347 (VICReg-style regulariser of an embedding table [n, d]): hinge on the per-dimension
    standard deviation, mean_d relu(gamma - sqrt(var_d + 1e-4)), and the squared off-diagonal entries of the
    covariance matrix summed and divided by d."""

    def __init__(self, name, loss_variance_weight, loss_covariance_weight, gamma=1):
        super().__init__()
        self.name = name
        self.loss_variance_weight = float(loss_variance_weight)
        self.loss_covariance_weight = float(loss_covariance_weight)
        self.gamma = gamma

    def forward(self, embeddings, lens=None):
        embs = _table(embeddings)
        n, d = embs.shape
        std_loss = torch.relu(self.gamma - torch.sqrt(embs.var(dim=0) + 1e-4)).mean()
        cen = embs - embs.mean(dim=0, keepdim=True)
        cov = (cen.t() @ cen) / (n - 1)
        off = cov - torch.diag(torch.diagonal(cov))
        cov_loss = off.pow(2).sum() / d
        return {f"loss_{self.name}_variance": (std_loss, self.loss_variance_weight),
                f"loss_{self.name}_covariance": (cov_loss, self.loss_covariance_weight)}


class AttributeMinCrossCovarianceRegLoss(nn.Module):
    """loss.py:252-296: the batch's two attribute vectors [B, d1], [B, d2], each centred on the mean of ITS
    embedding table (or of the batch when no table is given); mean squared entry of their [d1, d2]
    cross-covariance (over B - 1)."""

    def __init__(self, attr_name1, attr_name2, loss_cross_covariance_weight, gamma=1):
        super().__init__()
        self.attr_name1, self.attr_name2 = attr_name1, attr_name2
        self.loss_cross_covariance_weight = float(loss_cross_covariance_weight)

    def forward(self, batch_attr1, batch_attr2, attr1_embeddings, attr2_embeddings):
        t1 = _table(attr1_embeddings) if attr1_embeddings is not None else batch_attr1
        t2 = _table(attr2_embeddings) if attr2_embeddings is not None else batch_attr2
        a = batch_attr1 - t1.mean(dim=0, keepdim=True)
        b = batch_attr2 - t2.mean(dim=0, keepdim=True)
        cross = (a.t() @ b) / (batch_attr1.shape[0] - 1)
        loss = cross.pow(2).sum() / (t1.shape[1] * t2.shape[1])
        return {f"loss_{self.attr_name1}-{self.attr_name2}_cross_covariance": (loss, self.loss_cross_covariance_weight)}


def total_loss(loss_dict):
    """sum of value*weight (tts_lightning_modules.py:746-750)."""
    return sum(v * w for v, w in loss_dict.values())
