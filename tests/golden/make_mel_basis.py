#!/usr/bin/env python3
"""Fixture for the Slaney mel filterbank of the STFT front end (reference call site audio_processing.py:124-125:
librosa 0.8.0 `filters.mel(sr, n_fft, n_mels, fmin, fmax)`).  librosa is absent from the reference tree and from this image, so
the filterbank cannot be captured from the reference itself.  This script captures it from an INDEPENDENT implementation of
the same published algorithm that IS installed: Hugging Face transformers' `audio_utils.mel_filter_bank(norm="slaney",
mel_scale="slaney")` -- the function Whisper's feature extractor uses to reproduce librosa's filters, checked against librosa
in that project's own tests -- for the two geometries the reference's data configs use
(configs/RADMMM_LJS_22khz_data_config.yaml:19-25, configs/RADMMM_LJS_data_config.yaml:21-27).
The pin is therefore "a second, independently written and librosa-validated implementation agrees to float32 rounding", one
step short of librosa's own output; DESIGN.md section 2 says so.

    python tests/golden/make_mel_basis.py        -> tests/golden/mel_basis_hf.npz"""
import os

import numpy as np
import transformers
from transformers.audio_utils import mel_filter_bank

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    out = {"transformers_version": np.array(transformers.__version__)}
    for tag, sr in (("22k", 22050), ("16k", 16000)):
        n_fft, n_mels, fmin, fmax = 1024, 80, 0.0, 8000.0
        m = mel_filter_bank(num_frequency_bins=n_fft // 2 + 1, num_mel_filters=n_mels, min_frequency=fmin, max_frequency=fmax,
                            sampling_rate=sr, norm="slaney", mel_scale="slaney")
        out[f"mel_{tag}"] = np.ascontiguousarray(m.T).astype(np.float32)           # [n_mels, n_fft / 2 + 1], librosa's orientation
        out[f"args_{tag}"] = np.array([sr, n_fft, n_mels, fmin, fmax], dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "mel_basis_hf.npz"), **out)
    print({k: getattr(v, "shape", None) for k, v in out.items()})


if __name__ == "__main__":
    main()
