"""Stage-2 (``module2_mixed``) counterpart: the image-only refinement network over all 12 classes."""
