"""deepipr_amd -- MI355X-native (gfx950) implementation of DeepIPR's passport-layer hot path.

Host code is Python on PyTorch-ROCm and mirrors the reference's `models.layers`, `models.losses`,
`models.*_passport*` and `experiments.trainer*` interfaces; the passport arithmetic is hand-written HIP
behind the C ABI of include/deepipr_hip.h (deepipr_amd/csrc).  See DESIGN.md.
"""
__version__ = '0.1.0'
