"""MI355X-native STROTSS operator surface.

Same module names as the reference's `nn/` package (losses, model, strotss_utils, utils, rand) so
`run_strotss.py` and user code switch over by import path alone; every operator runs on the HIP
kernels of libstrotss_hip.so (include/strotss_hip.h) -- there is no CPU fallback.
"""
