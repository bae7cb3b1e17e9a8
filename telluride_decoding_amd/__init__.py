"""MI355X-native linear auditory-attention-decoding hot path.

Drop-in for the linear path of google/telluride_decoding (ridge TRF fit, CCA,
windowed correlation, attended-speaker decision) on hand-written HIP kernels for
gfx950 behind a C-ABI (include/td_hotpath.h).  There is no CPU fallback: the
compute entry points raise `HotPathUnavailable` without the HIP library + GPU.
"""
__version__ = '0.1.0'
