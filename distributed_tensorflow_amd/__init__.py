"""MI355X-native distributed data-parallel training framework."""
import os as _os

# The convolution library's algorithm search times every applicable solver once per new shape, its reference
# ("naive") kernels included: those take seconds per call on the large-batch ResNet stem (over a minute of the
# first training step) and are never the fastest. Keep them out of the search unless the user decides otherwise.
for _k in ("MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD", "MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_BWD",
           "MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_WRW"):
    _os.environ.setdefault(_k, "0")
del _k
