"""Host-side plumbing for the MI355X direct-force / surrogate hot path: the ctypes loader
of the C-ABI library (include/nbd.h), thin typed wrappers, input generators."""
