"""Prompt denoising shell: the STFT / chunking / overlap logic of the reference's UVR5 MDX-Net wrapper around a
pluggable separation network.  See ``mdx.py``."""
from .mdx import Inference, MDXConfig, UVR5  # noqa: F401
