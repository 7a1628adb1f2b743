"""MI355X-native depth engine for CNMNet's hot path."""
import os as _os

# Kernel arguments in DEVICE memory: a HIP runtime switch that is read when the runtime initialises (the first HIP call of the
# process), so it is set here, at import, unless the caller has chosen otherwise.  Every kernel begins by reading its arguments;
# from host-coherent memory that costs ~2 us before its first useful instruction (bench.py, round 5: +0.6 % frames/s, -2.3 us
# on the plane-sweep launch).
# It is a process-wide runtime setting (torch's kernels see it too): an embedding application that does not want a library to choose it
# sets CNM_KEEP_HIP_ENV=1 (nothing is touched then) or HIP_FORCE_DEV_KERNARG itself (setdefault never overrides).
if _os.environ.get("CNM_KEEP_HIP_ENV") != "1":
    _os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
