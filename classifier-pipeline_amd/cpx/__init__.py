"""cpx -- MI355X-native extract-and-classify hot path for CPTV thermal clips.

Host-side mirror of the reference's ``track`` / ``ml_tools`` / ``classify``
interfaces (TheCacophonyProject/classifier-pipeline) over hand-written HIP
kernels reached through the C-ABI in ``include/cpx.h`` (``libcpx_hip.so``).
There is no CPU fallback: the compute entry points raise if the HIP library or
a GPU is missing.
"""

__version__ = "0.1.0"
