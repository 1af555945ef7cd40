"""glow-tts_amd — MI355X-native (gfx950) Glow-TTS training hot path.

Host side (Python, mirroring the reference's module interface) over a C-ABI HIP library
(``include/glowtts_hip.h`` / ``libglowtts_hip.so`` built from ``csrc/*.hip``).  There is no
CPU or eager-PyTorch fallback for the kernels: if the HIP library is missing every op
raises.  Import as ``glow_tts_amd``.
"""
__version__ = "0.1.0"
