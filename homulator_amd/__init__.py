"""homulator_amd — MI355X-native execution backend for Homulator's FHE datapath (hmult / hrotate + hybrid
key switch).  Python here is plumbing only: a ctypes binding of the C ABI (include/homulator_hip.h) and the
multi-GPU launcher.  The compute path is the HIP library homulator_amd/lib/libhomulator_hip.so; there is no
CPU fallback, and importing `homulator_amd.hip` on a machine without the built library raises."""
__version__ = "0.1.0"

# Load order matters inside a Python process: PyTorch bundles its own HIP runtime (libamdhip64.so, soname
# libamdhip64.so.7) and RCCL.  If libhomulator_hip.so were loaded first it would pull /opt/rocm's copy and PyTorch would
# then map a SECOND runtime; streams and RCCL communicators must not cross runtimes ("unhandled cuda error" in
# ncclCommInitRank).  Importing torch first makes the dynamic loader resolve our dependency to the copy already mapped.
try:
    import torch as _torch  # noqa: F401
except ImportError:  # pure C/C++ hosts (the CLI) never come through here
    _torch = None
