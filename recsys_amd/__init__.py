"""recsys_amd: the CTR training hot path of wangruichens/recsys, MI355X-native.

Embedding gather + feature interaction + row-wise scatter + TF-1 Adam as hand-written gfx950 HIP
kernels behind a C ABI (include/rsx.h, recsys_amd/csrc), with a PyTorch-ROCm host that mirrors the
reference's Estimator surface (model_fn / input_fn / Estimator).  See DESIGN.md.
"""
__version__ = "0.1.0"
