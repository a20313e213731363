"""Host-side mirror of the reference's ``myQL`` package (quantisation framework) whose mode-1
graph lowers to one fused device call (libsesrq.so)."""
