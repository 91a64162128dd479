"""MI355X-native implementation of DeMethify's deconvolution solver hot path.

Host-side mirror of the reference's interface (``deconvolution``, ``bootstrap``, ``ic``,
``init_func``, ``demethify`` CLI) over hand-written HIP kernels behind a C-ABI
(include/demethify_hip.h, demethify_amd/csrc/).  Importing the package does not touch the GPU.
"""
__version__ = "0.1.0"
