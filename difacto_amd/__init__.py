"""difacto_amd — MI355X-native FM/SGD worker path behind dmlc/difacto's interfaces.

The product is the C-ABI library ``libdifacto_hip.so`` (include/difacto_hip.h);
this package is the ctypes binding used by tests, bench.py and the multi-GPU
driver.  There is no CPU fallback: importing :mod:`difacto_amd.capi` without the
built library, or creating a context without a HIP device, raises.
"""
__all__ = ["capi", "build", "synth", "sharded"]
