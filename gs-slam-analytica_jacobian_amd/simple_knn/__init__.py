"""Overlay of the reference's `simple_knn` package (submodules/simple-knn): only `_C.distCUDA2` exists upstream."""
