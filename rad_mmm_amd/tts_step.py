"""The training step's caller: the glue of TTSModel.training_step
(tts_lightning_modules.py:643-750) without Lightning -- unpack the batch, embed speaker / accent /
text, run the text encoder, the alignment attention (+ monotonic alignment search once
global_step >= binarization_start_iter), build the decoder context, call the flow decoder and
aggregate the loss dictionary.  Module attribute names follow the reference (`text_embeddings`,
`text_encoder`, `speaker_embeddings`, `accent_embeddings`, `attention`, `decoder`, ...), so a
TTSModel state_dict loads key for key.  Everything runs on the device: the reference's per-item
MAS on the CPU (`.cpu().numpy()`, :270-284) is one batched HIP launch."""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch
from torch import nn

from .alignment import binarize_attention
from .attention import ConvAttention
from .common import SequenceLength, get_mask_from_lengths


class TTSTrainingStep(nn.Module):
    def __init__(self, text_encoder: nn.Module, decoder: nn.Module, decoder_loss: nn.Module, n_speakers=1, n_accents=1,
                 n_text_tokens=185, n_text_dim=512, n_speaker_dim=16, n_accent_dim=8, n_mel_channels=80, use_accent=True,
                 use_accent_emb_for_encoder=False, use_accent_emb_for_decoder=False, use_accent_emb_for_alignment=False,
                 use_speaker_emb_for_alignment=False, binarization_start_iter=20000, f0_predictor=None,
                 f0_predictor_loss=None, energy_predictor=None, energy_predictor_loss=None, voiced_predictor=None,
                 voiced_predictor_loss=None, duration_predictor=None, duration_predictor_loss=None,
                 f0_loss_voiced_only=True, speaker_embed_regularization_loss=None, accent_embed_regularization_loss=None,
                 speaker_accent_cross_regularization_loss=None):
        super().__init__()
        self.text_embeddings = nn.Embedding(n_text_tokens, n_text_dim)
        self.text_encoder = text_encoder
        self.speaker_embeddings = nn.Embedding(n_speakers, n_speaker_dim)
        self.use_accent = bool(use_accent)
        if self.use_accent:
            self.accent_embeddings = nn.Embedding(n_accents, n_accent_dim)
        self.use_accent_emb_for_encoder = bool(use_accent_emb_for_encoder)
        self.use_accent_emb_for_decoder = bool(use_accent_emb_for_decoder)
        self.use_accent_emb_for_alignment = bool(use_accent_emb_for_alignment)
        self.use_speaker_emb_for_alignment = bool(use_speaker_emb_for_alignment)
        key_dim = n_text_dim + (n_accent_dim if use_accent_emb_for_alignment else
                                (n_speaker_dim if use_speaker_emb_for_alignment else 0))
        self.attention = ConvAttention(n_mel_channels, key_dim)
        self.decoder = decoder
        decoder_loss.n_group_size = decoder.n_group_size          # tts_lightning_modules.py:153
        self.decoder_criterion = decoder_loss
        self.binarization_start_iter = binarization_start_iter
        self.f0_loss_voiced_only = f0_loss_voiced_only
        self.f0_predictor, self.f0_predictor_loss = f0_predictor, f0_predictor_loss
        self.energy_predictor, self.energy_predictor_loss = energy_predictor, energy_predictor_loss
        self.voiced_predictor, self.voiced_predictor_loss = voiced_predictor, voiced_predictor_loss
        self.duration_predictor, self.duration_predictor_loss = duration_predictor, duration_predictor_loss
        # embedding regularisers (tts_lightning_modules.py:187-201; configs/RADMMM_model_config.yaml:49-61)
        self.speaker_embed_regularization_loss = speaker_embed_regularization_loss
        self.accent_embed_regularization_loss = accent_embed_regularization_loss
        self.speaker_accent_cross_regularization_loss = speaker_accent_cross_regularization_loss
        self.binarize = False                                    # validation uses the flag training last set (:644-646)
        # mean-reduce the step's loss terms across ranks for logging (Lightning's `sync_dist=True`): outputs["losses_global"]
        # is a handle whose wait() returns {name: global mean}; a no-op without a process group
        self.sync_dist = True

    # ---- tts_lightning_modules.py:543-545, 246-268 ---------------------------------------------
    @staticmethod
    def mel_scale(mel):
        return (mel + 5) / 2

    def encode_speaker(self, spk_ids):
        return self.speaker_embeddings(spk_ids)

    def encode_accent(self, accent_ids):
        return self.accent_embeddings(accent_ids)

    def encode_text(self, text, in_lens, accent_vecs=None, max_len=None):
        emb = self.text_embeddings(text).transpose(1, 2)
        x = emb
        if accent_vecs is not None:
            x = torch.cat((emb, accent_vecs[..., None].expand(-1, -1, emb.shape[-1])), 1)
        from .encoder import Encoder as _HipEncoder
        enc = self.text_encoder.infer(x) if in_lens is None else (
            self.text_encoder(x, in_lens, max_len) if (max_len is not None and isinstance(self.text_encoder, _HipEncoder))
            else self.text_encoder(x, in_lens))
        return enc.transpose(1, 2), emb

    # ---- tts_lightning_modules.py:440-475 ------------------------------------------------------
    def compute_attention(self, mel, txt_emb, spk_vecs, accent_vecs, out_lens, in_lens, attn_prior, binarize=False,
                          max_in_len=None):
        attn_mask = get_mask_from_lengths(in_lens, max_in_len)[..., None] == 0
        keys = txt_emb
        if self.use_accent_emb_for_alignment:
            keys = torch.cat((keys, accent_vecs[:, :, None].expand(-1, -1, txt_emb.shape[2]).detach()), 1)
        elif self.use_speaker_emb_for_alignment:
            keys = torch.cat((keys, spk_vecs[:, :, None].expand(-1, -1, txt_emb.shape[2]).detach()), 1)
        attn_soft, attn_logprob = self.attention(mel, keys, out_lens, attn_mask, key_lens=in_lens, attn_prior=attn_prior)
        attn_hard = None
        if binarize:
            attn = binarize_attention(attn_soft, in_lens, out_lens)
            attn_hard = attn_soft + (attn - attn_soft).detach()
        else:
            attn = attn_soft
        return attn, attn_soft, attn_hard, attn_logprob

    # ---- tts_lightning_modules.py:643-750 ------------------------------------------------------
    def training_step(self, batch: Dict[str, torch.Tensor], global_step: int = 0
                      ) -> Tuple[torch.Tensor, Dict[str, Tuple[torch.Tensor, float]], Dict[str, torch.Tensor]]:
        """batch keys as data.py:756-788 (the ones this step reads): mel, speaker_ids, accent_ids, text,
        input_lengths, output_lengths, attn_prior, f0, energy_avg (+ voiced_mask, speaker_f0_mean/std when
        the attribute predictors are attached).  Returns (loss, {name: (value, weight)}, outputs)."""
        binarize = self.binarize = global_step >= self.binarization_start_iter
        # (the batch may carry host copies of the lengths -- the collate function has them before the batch moves to the
        #  device -- as "input_lengths_host" / "output_lengths_host": no device -> host read, i.e. no synchronisation, then)
        in_lens = SequenceLength(batch["input_lengths"], batch.get("input_lengths_host"))
        out_lens = SequenceLength(batch["output_lengths"], batch.get("output_lengths_host"))
        max_in = int(in_lens.lengths_host.max())
        mel = self.mel_scale(batch["mel"])
        spk_vecs = self.encode_speaker(batch["speaker_ids"])
        accent_vecs = self.encode_accent(batch["accent_ids"]) if self.use_accent else None
        txt_enc, txt_emb = self.encode_text(batch["text"], in_lens.lengths,
                                            accent_vecs if self.use_accent_emb_for_encoder else None, max_in)
        attn, attn_soft, _, attn_logprob = self.compute_attention(
            mel, txt_emb, spk_vecs, accent_vecs, out_lens.lengths, in_lens.lengths, batch["attn_prior"], binarize, max_in)
        context = torch.bmm(txt_enc, attn.squeeze(1).transpose(1, 2))
        f0, energy_avg = batch.get("f0"), batch.get("energy_avg")
        outputs = self.decoder(mel, spk_vecs, context, out_lens, f0=f0, energy_avg=energy_avg, accent_vecs=accent_vecs)
        outputs.update(attn=attn, attn_soft=attn_soft, attn_logprob=attn_logprob, context=context, spk_vecs=spk_vecs,
                       accent_vecs=accent_vecs)
        losses: Dict[str, Tuple[torch.Tensor, float]] = {}
        if self.decoder.training:
            losses.update(self.decoder_criterion(outputs, in_lens, out_lens, global_step))
        acc_d = accent_vecs.detach() if accent_vecs is not None else None
        # the mel-rate predictors (f0, energy, voiced: tts_lightning_modules.py:688-717) read the same detached context over
        # the same frames: their calls are collected and run through attribute_predictors.dap_forward_many, which merges
        # their bi-LSTMs into one recurrence when they are this package's ConvLSTMLinearDAP of one shape
        mel_rate = []
        if self.f0_predictor is not None:
            mel_rate.append(("f0", self.f0_predictor, ((f0.unsqueeze(1), context.detach(), spk_vecs.detach(), out_lens,
                                                        batch.get("speaker_f0_mean"), batch.get("speaker_f0_std"), acc_d), {})))
        if self.energy_predictor is not None:
            mel_rate.append(("energy", self.energy_predictor, ((energy_avg.unsqueeze(1), context.detach(), spk_vecs.detach(), out_lens),
                                                               {"accent_emb": acc_d})))
        if self.voiced_predictor is not None:
            mel_rate.append(("voiced", self.voiced_predictor, ((batch["voiced_mask"].unsqueeze(1), context.detach(), spk_vecs.detach(),
                                                                out_lens), {"accent_emb": acc_d})))
        from .attribute_predictors import ConvLSTMLinearDAP, dap_forward_many
        if len(mel_rate) > 1 and all(isinstance(m[1], ConvLSTMLinearDAP) for m in mel_rate):
            outs = dap_forward_many([m[1] for m in mel_rate], [m[2] for m in mel_rate])
        else:
            outs = [m[1](*m[2][0], **m[2][1]) for m in mel_rate]
        for (name, _, _), o in zip(mel_rate, outs):
            if name == "f0":
                m = batch["voiced_mask"].unsqueeze(1) if self.f0_loss_voiced_only else None
                losses.update(self.f0_predictor_loss(o, in_lens, out_lens, global_step, mask=m))
            elif name == "energy":
                losses.update(self.energy_predictor_loss(o, in_lens, out_lens, global_step))
            else:
                losses.update(self.voiced_predictor_loss(o, in_lens, out_lens, global_step))
        if self.duration_predictor is not None:
            o = self.duration_predictor(attn.sum(2).detach(), txt_enc.detach(), spk_vecs.detach(), in_lens, accent_emb=acc_d)
            losses.update(self.duration_predictor_loss(o, None, None, global_step, in_lens.mask.unsqueeze(1)))
        losses.update(self._embedding_regularisers(spk_vecs, accent_vecs))
        loss = None
        for v, w in losses.values():
            loss = v * w if loss is None else loss + v * w
        if self.sync_dist:
            # the reference logs every term with sync_dist=True (tts_lightning_modules.py:746-749: a mean all-reduce per term and
            # step); here: one coalesced collective behind the forward pass, waited for by whoever logs (ddp.reduce_loss_dict)
            from .ddp import reduce_loss_dict
            outputs["losses_global"] = reduce_loss_dict(dict(losses, loss=(loss, 1.0)))
        return loss, losses, outputs

    def _embedding_regularisers(self, spk_vecs, accent_vecs):
        """tts_lightning_modules.py:729-745 (and :837-853 in validation)"""
        out = {}
        if self.speaker_embed_regularization_loss is not None:
            out.update(self.speaker_embed_regularization_loss(self.speaker_embeddings))
        if self.accent_embed_regularization_loss is not None:
            out.update(self.accent_embed_regularization_loss(self.accent_embeddings))
        if self.speaker_accent_cross_regularization_loss is not None:
            out.update(self.speaker_accent_cross_regularization_loss(spk_vecs, accent_vecs, self.speaker_embeddings,
                                                                     self.accent_embeddings))
        return out

    @torch.no_grad()
    def validation_step(self, batch: Dict[str, torch.Tensor], global_step: int = 0
                        ) -> Tuple[torch.Tensor, Dict[str, Tuple[torch.Tensor, float]], Dict[str, torch.Tensor]]:
        """tts_lightning_modules.py:752-860: the same pass without gradients; the decoder criterion is evaluated
        at step 100000 (all loss terms on), alignments are binarized when training last was (`self.binarize`),
        and the predictors see the un-detached context.  `global_step` only reaches the predictor losses."""
        # (the batch may carry host copies of the lengths -- the collate function has them before the batch moves to the
        #  device -- as "input_lengths_host" / "output_lengths_host": no device -> host read, i.e. no synchronisation, then)
        in_lens = SequenceLength(batch["input_lengths"], batch.get("input_lengths_host"))
        out_lens = SequenceLength(batch["output_lengths"], batch.get("output_lengths_host"))
        max_in = int(in_lens.lengths_host.max())
        mel = self.mel_scale(batch["mel"])
        spk_vecs = self.encode_speaker(batch["speaker_ids"])
        accent_vecs = self.encode_accent(batch["accent_ids"]) if self.use_accent else None
        txt_enc, txt_emb = self.encode_text(batch["text"], in_lens.lengths,
                                            accent_vecs if self.use_accent_emb_for_encoder else None, max_in)
        attn, attn_soft, _, attn_logprob = self.compute_attention(
            mel, txt_emb, spk_vecs, accent_vecs, out_lens.lengths, in_lens.lengths, batch["attn_prior"], self.binarize, max_in)
        context = torch.bmm(txt_enc, attn.squeeze(1).transpose(1, 2))
        f0, energy_avg = batch.get("f0"), batch.get("energy_avg")
        outputs = self.decoder(mel, spk_vecs, context, out_lens, f0=f0, energy_avg=energy_avg, accent_vecs=accent_vecs)
        outputs.update(attn=attn, attn_soft=attn_soft, attn_logprob=attn_logprob, context=context, spk_vecs=spk_vecs,
                       accent_vecs=accent_vecs, txt_enc=txt_enc)
        losses = dict(self.decoder_criterion(outputs, in_lens, out_lens, 100000))
        if self.f0_predictor is not None:
            o = self.f0_predictor(f0.unsqueeze(1), context, spk_vecs, out_lens, batch.get("speaker_f0_mean"),
                                  batch.get("speaker_f0_std"), accent_vecs)
            m = batch["voiced_mask"].unsqueeze(1) if self.f0_loss_voiced_only else None
            losses.update(self.f0_predictor_loss(o, in_lens, out_lens, global_step, mask=m))
            outputs["f0_outputs"] = o
        if self.energy_predictor is not None:
            o = self.energy_predictor(energy_avg.unsqueeze(1), context, spk_vecs, out_lens, accent_emb=accent_vecs)
            losses.update(self.energy_predictor_loss(o, in_lens, out_lens, global_step))
            outputs["energy_outputs"] = o
        if self.voiced_predictor is not None:
            o = self.voiced_predictor(batch["voiced_mask"].unsqueeze(1), context, spk_vecs, out_lens, accent_emb=accent_vecs)
            losses.update(self.voiced_predictor_loss(o, in_lens, out_lens, global_step))
            outputs["voiced_outputs"] = o
        if self.duration_predictor is not None:
            o = self.duration_predictor(attn.sum(2), txt_enc, spk_vecs, in_lens, accent_emb=accent_vecs)
            losses.update(self.duration_predictor_loss(o, None, None, global_step, in_lens.mask.unsqueeze(1)))
            outputs["duration_outputs"] = o
        losses.update(self._embedding_regularisers(spk_vecs, accent_vecs))
        loss = None
        for v, w in losses.values():
            loss = v * w if loss is None else loss + v * w
        return loss, losses, outputs
