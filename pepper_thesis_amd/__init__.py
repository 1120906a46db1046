"""pepper_thesis_amd — MI355X-native pileup summary-image builder and RNN inference for PEPPER.

Host-side mirror of the two reference operators on the hot path above the
C-ABI of include/pepper_hip.h; the compute lives in hand-written HIP kernels under csrc/.
"""
__version__ = "0.1.0"
