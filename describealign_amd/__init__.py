"""describealign_amd -- MI355X-native alignment core for describealign's hot path.

Host code mirrors the reference's interface (get_energy, get_zero_crossings, get_freq_bands,
align, combine, plot_alignment); compute runs in hand-written HIP kernels for gfx950 behind
the C ABI in include/dalign.h (libdalign.so).  There is no CPU fallback.
"""
__version__ = "0.1.0"
REFERENCE_VERSION = "2.0.8"   # julbean/describealign version whose results are reproduced
