"""mmidet_hip: host side of the MI355X-native MMI-Det two-stream hot path.

Python on PyTorch-ROCm for tensors/streams/autograd glue; all device work goes through the C ABI of
``csrc/libmmidet_hip.so`` (include/mmidet_hip.h).  There is NO CPU or ATen fallback: importing ``mmidet_hip.lib``
without the built library raises.
"""
