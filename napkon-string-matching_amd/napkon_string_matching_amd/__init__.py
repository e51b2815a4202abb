"""MI355X-native all-pairs similarity scorer: a drop-in for the cross-cohort match loop of
BIH-CEI/napkon-string-matching (``napkon_string_matching.matching`` / ``compare``).

The per-pair arithmetic runs in hand-written HIP kernels (``csrc/``, built into
``csrc/libnsm_hip.so``) behind a C ABI (``include/nsm_hip.h``).  This Python package is the host
side: it keeps the reference's plugin / ``ComparableData`` / ``Matcher`` surface, encodes the
per-item operands into HBM tables and calls the library through ctypes.  There is no CPU
fallback: importing works anywhere, scoring raises if the library or the GPU is missing.
"""
__version__ = "0.1.0"
