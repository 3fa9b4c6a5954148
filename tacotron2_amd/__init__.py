"""MI355X-native mel-spectrogram -> waveform vocoding (HiFi-GAN V1 generator, WaveGlow inference)
behind the call signatures of gerard-bailly/tacotron2's `hifigan.models.Generator` and
`glow.WaveGlow`.  The arithmetic lives in libmelvoc.so (hand-written HIP for gfx950, C ABI in
include/melvoc.h); this package is the host-side mirror of the reference's Python interface."""
__all__ = ["hifigan", "waveglow", "synth", "vocoder", "batch"]
