"""svx -- MI355X (gfx950) implementation of Speech-Vecalign's segment-alignment hot path.

Mirrors the reference's module layout for the path (svecalign.vecalign.{dp_core,dp_utils,vecalign,
score}, svecalign.seg_align.align, svecalign.utils.{embedding_utils,file_utils}); all numerics run
in hand-written HIP kernels behind the C ABI of include/svx.h (libsvx.so).  There is no CPU
fallback: without the library or a GPU the compute entry points raise.
"""
__version__ = "0.1"
