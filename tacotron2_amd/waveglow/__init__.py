"""Drop-in for the reference's `waveglow/` folder: put this directory on sys.path (as do_syn.py:147
does with 'waveglow') so pickled checkpoints resolve `glow.WaveGlow` to the native class."""
import os

from .glow import WN, Invertible1x1Conv, WaveGlow  # noqa: F401

GLOW_PATH = os.path.dirname(os.path.abspath(__file__))
