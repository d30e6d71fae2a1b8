"""Drop-in mirror of the reference's `models` package for the hot path:
same import paths, constructor kwargs, forward signatures and state_dict keys; the arithmetic runs
in libhvc_hip.so (hand-written gfx950 kernels) instead of stock ATen ops."""
