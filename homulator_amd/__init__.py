"""homulator_amd — MI355X-native execution backend for Homulator's FHE datapath (hmult / hrotate + hybrid
key switch).  Python here is plumbing only: a ctypes binding of the C ABI (include/homulator_hip.h) and the
multi-GPU launcher.  The compute path is the HIP library homulator_amd/lib/libhomulator_hip.so; there is no
CPU fallback, and importing `homulator_amd.hip` on a machine without the built library raises."""
__version__ = "0.1.0"
