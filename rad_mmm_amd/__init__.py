"""rad_mmm_amd -- MI355X-native (gfx950) hot path of NVIDIA/RAD-MMM: the normalizing-flow
mel-decoder training step (forward + NLL + backward), its alignment attention / MAS and the
STFT->mel front end, behind the reference's decoder/loss plug-in API.

Importing the package loads libradmmm_hip.so (built in-tree by `__graft_entry__.build()`);
there is no CPU or eager-PyTorch fallback: without the library the import fails.

Drop-in points (reference class -> replacement):
    decoders.RADMMMFlow        -> rad_mmm_amd.decoders.RADMMMFlow
    loss.RADMMMLoss/RADTTSLoss -> rad_mmm_amd.loss.RADMMMLoss / RADTTSLoss
    common.SequenceLength      -> rad_mmm_amd.common.SequenceLength
    common.ConvAttention       -> rad_mmm_amd.attention.ConvAttention
    alignment.mas_width1       -> rad_mmm_amd.alignment.mas_width1 / binarize_attention
    audio_processing.TacotronSTFT -> rad_mmm_amd.audio_processing.TacotronSTFT
"""
from . import _lib  # noqa: F401  (fails loudly if the HIP library is missing)

__version__ = "0.1.0"
