"""Host-side mirror of the reference's ``depthnet`` package: same module names, class
names, constructor arguments, forward signatures, return structure and state_dict keys
(reference depthnet/depthNet_model.py, depth_util.py, inverse_warp.py), computing on the
MI355X through libcnm_engine.so.  ``import cnmnet_amd.depthnet as depthnet`` is the
intended drop-in."""
from .depthNet_model import depthNet, DepthRefineNet          # noqa: F401
from .depth_util import Depth2normal, get_normal_by_planes, process_camera_parameters, get_pixel_coordinates  # noqa: F401
from .inverse_warp import inverse_warp, pixel2cam              # noqa: F401
from .losses import IdepthLoss, IdepthLoss_234, IdepthwithProbLoss, surface_normal_loss  # noqa: F401
