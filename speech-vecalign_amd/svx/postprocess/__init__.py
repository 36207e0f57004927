"""Pure-Python alignment post-filters of the reference's svecalign/postprocess/ that sit directly behind the
aligner: cost filter, consecutive-alignment concatenation, duration filter."""
