"""Prompt denoiser: the reference's UVR5 MDX-Net wrapper -- STFT / chunking / overlap shell (``mdx.py``) around the
ConvTDFNet separation network on the HIP engine (``lemas_mdx_*``; weights via ``onnx_weights.py``)."""
from .mdx import Inference, MDXConfig, UVR5, resolve_model_dir  # noqa: F401
