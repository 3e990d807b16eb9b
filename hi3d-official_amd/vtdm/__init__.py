"""MI355X mirror of the reference's thin Hi3D glue package `vtdm` (inference surface only)."""
