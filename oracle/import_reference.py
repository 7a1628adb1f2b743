"""Import the UNMODIFIED reference (xxlong0/CNMNet) from /root/reference on CPU.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Works only in the authoring
container: /root/reference does not exist on the GPU box, and nothing executed
there may call this module.

The reference hard-codes CUDA tensor types and imports cv2 / tensorflow at
module scope although the hot path never uses them:
  * depthnet/depth_util.py:6-7, depthNet_model.py:10, scannet/preprocess.py:8,13
    -> stub modules in sys.modules
  * torch.cuda.FloatTensor (depth_util.py:20,53-54, depthNet_model.py:199,204)
    -> aliased to torch.FloatTensor
  * Tensor.get_device() fed to .to() (inverse_warp.py:36,102, depth_util.py:158)
    -> returns the tensor's device object instead of -1
No file under /root/reference is edited or copied.
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("CNM_REFERENCE_ROOT", "/root/reference")


def available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "depthnet", "depthNet_model.py"))


_loaded = None


def load():
    """Returns a namespace with depthNet, DepthRefineNet, Depth2normal,
    inverse_warp, pixel2cam, process_camera_parameters, get_pixel_coordinates, losses."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not available():
        raise RuntimeError("reference checkout not present at %s" % REFERENCE_ROOT)
    import torch

    for name in ("cv2", "tensorflow", "tensorflow.python", "tensorflow.python.lib",
                 "tensorflow.python.lib.io", "tensorflow.python.lib.io.file_io"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    torch.cuda.FloatTensor = torch.FloatTensor
    torch.Tensor.get_device = lambda self: self.device
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    # our own package also has a sub-package called "depthnet"; the reference's
    # top-level "depthnet" is a different module name, so no clash.
    import depthnet.depthNet_model as m
    import depthnet.depth_util as u
    import depthnet.inverse_warp as iw
    import depthnet.losses as ls

    ns = types.SimpleNamespace(
        depthNet=m.depthNet, DepthRefineNet=m.DepthRefineNet,
        Depth2normal=u.Depth2normal, get_normal_by_planes=u.get_normal_by_planes,
        process_camera_parameters=u.process_camera_parameters,
        get_pixel_coordinates=u.get_pixel_coordinates,
        inverse_warp=iw.inverse_warp, pixel2cam=iw.pixel2cam, iw_module=iw,
        losses=ls)
    _loaded = ns
    return ns
