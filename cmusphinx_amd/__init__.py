"""cmusphinx_amd -- MI355X-native GMM senone scoring + HMM Viterbi backend for
CMU Sphinx-3, behind the C ABI of include/cmusphinx_amd.h.

This Python package is only the test / bench harness around
``libcmusphinx_amd.so`` (host C + hand-written HIP for gfx950).  There is no
Python or CPU fallback: if the shared library is missing, import of
:mod:`cmusphinx_amd.lib` fails loudly, and every scoring call fails with
``S3A_ENODEV`` when no GPU is present.
"""
__all__ = ["s3io", "synth", "lib"]
