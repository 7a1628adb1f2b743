"""ctypes binding of include/cnm_engine.h (the C ABI of libcnm_engine.so).

There is deliberately no fallback: if the library is missing the import of any
operator fails loudly -- this package never computes on the CPU and never routes
through oracle/.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CNM_ENGINE_LIB") or os.path.join(_HERE, "lib", "libcnm_engine.so")   # CNM_ENGINE_LIB: an A/B build of the same ABI (bench.py's unit-order pass)

c_fp = C.c_void_p          # device pointers travel as integers
c_i, c_f, c_d, c_ll, c_sz = C.c_int, C.c_float, C.c_double, C.c_longlong, C.c_size_t


class LayerInfo(C.Structure):
    _fields_ = [("conv_key", C.c_char_p), ("bn_key", C.c_char_p), ("Cin", c_i), ("Cout", c_i),
                ("ksize", c_i), ("stride", c_i), ("rot", c_i), ("is_head", c_i)]


class LayerWeights(C.Structure):
    _fields_ = [("w", c_fp), ("b", c_fp), ("u", c_fp), ("u4", c_fp), ("uu", c_fp), ("bu", c_fp), ("wr", c_fp), ("u4q", c_fp)]


# name -> (restype, argtypes); mirrors include/cnm_engine.h declaration by declaration
PROTOTYPES = {
    "cnm_abi_version": (c_i, []),
    "cnm_status_string": (C.c_char_p, [c_i]),
    "cnm_homography_terms_f32": (c_i, [c_fp, c_fp, c_fp, c_i, c_i, c_fp]),
    "cnm_idepth_range_host": (c_i, [c_d, C.POINTER(c_d), C.POINTER(c_d)]),
    "cnm_planesweep_workspace_floats": (c_sz, [c_i, c_i, c_i, c_i]),
    "cnm_planesweep_volume_nchw_f32": (c_i, [c_fp, c_fp, c_fp, c_fp, c_fp, c_sz, c_i, c_i, c_i, c_i, c_i, c_d, c_d, c_fp]),
    "cnm_planesweep_cat_c4_f32": (c_i, [c_fp, c_fp, c_fp, c_fp, c_fp, c_sz, c_i, c_i, c_i, c_i, c_i, c_d, c_d, c_fp]),
    "cnm_packed_conv_floats": (c_sz, [c_i, c_i, c_i]),
    "cnm_pack_conv_bn_f32": (c_i, [c_fp] * 6 + [c_f, c_i, c_i, c_i, c_i, c_fp, c_fp, c_fp]),
    "cnm_conv2d_c4_f32": (c_i, [c_fp, c_i, c_i, c_i, c_fp, c_i, c_i, c_i, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_fp]),
    "cnm_conv2d_cat2_c4_f32": (c_i, [c_fp, c_i, c_i, c_i, c_fp, c_i, c_i, c_i, c_fp, c_i, c_i, c_i, c_fp, c_fp,
                                     c_i, c_i, c_i, c_i, c_i, c_i, c_fp]),
    "cnm_packed_winograd_floats": (c_sz, [c_i, c_i]),
    "cnm_pack_winograd_bn_f32": (c_i, [c_fp, c_fp, c_fp, c_f, c_i, c_i, c_i, c_fp, c_fp]),
    "cnm_conv3x3_winograd_c4_f32": (c_i, [c_fp, c_i, c_i, c_i, c_fp, c_i, c_i, c_i, c_fp, c_i, c_i, c_i, c_fp, c_fp,
                                          c_i, c_i, c_i, c_i, c_fp]),
    "cnm_engine_status": (c_i, [c_i]),
    "cnm_debug_sync_generation": (c_i, [c_fp, c_i, c_fp]),
    "cnm_tune_sync_spin_limit": (C.c_uint, [C.c_uint]),
    "cnm_tune_wino4_min_workgroups": (c_i, [c_i]),
    "cnm_tune_refine_side_stream": (c_i, [c_i]),
    "cnm_tune_upsampled_min_pixels": (c_i, [c_i]),
    "cnm_tune_upsampled_min_pixels_f16": (c_i, [c_i]),
    "cnm_tune_glds_tile": (c_i, [c_i]),
    "cnm_tune_wino36_staged": (c_i, [c_i]),
    "cnm_tune_rows_wide": (c_i, [c_i]),
    "cnm_tune_wino4_small": (c_i, [c_i]),
    "cnm_packed_winograd4_floats": (c_sz, [c_i, c_i]),
    "cnm_pack_winograd4_bn_f32": (c_i, [c_fp, c_fp, c_fp, c_f, c_i, c_i, c_i, c_fp, c_fp]),
    "cnm_pack_winograd4_dgrad_f32": (c_i, [c_fp, c_i, c_i, c_i, c_fp, c_fp]),
    "cnm_pack_winograd4_batch_f32": (c_i, [c_fp, c_i, c_i, c_fp]),
    "cnm_conv3x3_winograd4_c4_f32": (c_i, [c_fp, c_i, c_i, c_i, c_fp, c_i, c_i, c_i, c_fp, c_i, c_i, c_i, c_fp, c_fp,
                                           c_i, c_i, c_i, c_i, c_fp]),
    "cnm_conv3x3_s2_winograd_c4_f32": (c_i, [c_fp, c_i, c_i, c_i, c_fp, c_i, c_i, c_i, c_fp, c_fp, c_i, c_i, c_i, c_i, c_fp]),
    "cnm_conv3x3_upsampled_winograd4_c4_f32": (c_i, [c_fp, c_i, c_i, c_i, c_fp, c_i, c_i, c_i, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_fp]),
    "cnm_wino36_sync_floats": (c_sz, []),
    "cnm_packed_winograd4_quad_floats": (c_sz, [c_i, c_i]),
    "cnm_repack_winograd4_quad_f32": (c_i, [c_fp, c_i, c_i, c_fp, c_fp]),
    "cnm_conv3x3_winograd4q_ok": (c_i, [c_i, c_i, c_i]),
    "cnm_conv3x3_winograd4q_sync_c4_f32": (c_i, [c_fp, c_i, c_i, c_i, c_fp, c_i, c_i, c_i, c_fp, c_i, c_i, c_i, c_fp, c_fp,
                                                 c_i, c_i, c_i, c_i, c_fp, c_sz, c_fp]),
    "cnm_tune_wino36_quad": (c_i, [c_i]),
    "cnm_tune_wgrad_streamk": (c_i, [c_i]),
    "cnm_tune_wgrad_streamk_share": (c_i, [c_i]),
    "cnm_tune_wgrad_linear": (c_i, [c_i]),
    "cnm_conv3x3_winograd4_sync_c4_f32": (c_i, [c_fp, c_i, c_i, c_i, c_fp, c_i, c_i, c_i, c_fp, c_i, c_i, c_i, c_fp, c_fp,
                                                c_i, c_i, c_i, c_i, c_fp, c_sz, c_fp]),
    "cnm_conv5x5_winograd_sync_c4_f32": (c_i, [c_fp, c_i, c_i, c_i, c_fp, c_i, c_i, c_i, c_fp, c_i, c_i, c_i, c_fp, c_fp,
                                               c_i, c_i, c_i, c_i, c_fp, c_sz, c_fp]),
    "cnm_conv3x3_upsampled_winograd4_sync_c4_f32": (c_i, [c_fp, c_i, c_i, c_i, c_fp, c_i, c_i, c_i, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_fp, c_sz, c_fp]),
    "cnm_conv3x3_phase_scatter_winograd4_sync_c4_f32": (c_i, [c_fp, c_i, c_i, c_i, c_fp, c_i, c_i, c_i, c_fp, c_fp, c_i, c_i, c_i, c_i, c_fp, c_sz, c_fp]),
    "cnm_conv4x4_phase_scatter_winograd_sync_c4_f32": (c_i, [c_fp, c_i, c_i, c_i, c_fp, c_i, c_i, c_i, c_fp, c_fp, c_i, c_i, c_i, c_i, c_fp, c_sz, c_fp]),
    "cnm_pack_winograd36_f32": (c_i, [c_fp, c_i, c_i, c_i, c_fp, c_fp]),
    "cnm_packed_winograd4_s2_floats": (c_sz, [c_i, c_i]),
    "cnm_pack_winograd4_s2_bn_f32": (c_i, [c_fp, c_fp, c_fp, c_f, c_i, c_i, c_i, c_i, c_fp, c_fp]),
    "cnm_conv_s2_winograd4_ok": (c_i, [c_i, c_i, c_i, c_i]),
    "cnm_conv_s2_winograd4_sync_c4_f32": (c_i, [c_fp, c_i, c_i, c_i, c_fp, c_i, c_i, c_i, c_fp, c_i, c_i, c_i, c_fp, c_fp,
                                                c_i, c_i, c_i, c_i, c_i, c_fp, c_sz, c_fp]),
    "cnm_tune_wino4_s2": (c_i, [c_i]),
    "cnm_packed_upsampled_ring_floats": (c_sz, [c_i, c_i]),
    "cnm_pack_upsampled_ring_f32": (c_i, [c_fp, c_fp, c_fp, c_f, c_i, c_i, c_fp, c_fp]),
    "cnm_conv3x3_upsampled_ring_c4_f32": (c_i, [c_fp, c_i, c_i, c_i, c_fp, c_i, c_i, c_i, c_fp, c_fp, c_i, c_i, c_i, c_i, c_fp]),
    "cnm_conv3x3_upsampled_c8_f16": (c_i, [c_fp, c_i, c_i, c_i, c_fp, c_i, c_i, c_i, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_fp]),
    "cnm_conv3x3_upsampled_ring_c8_f16": (c_i, [c_fp, c_i, c_i, c_i, c_fp, c_i, c_i, c_i, c_fp, c_fp, c_i, c_i, c_i, c_i, c_fp]),
    "cnm_pack_winograd5x5_bn_f32": (c_i, [c_fp, c_fp, c_fp, c_f, c_i, c_i, c_i, c_fp, c_fp]),
    "cnm_conv5x5_winograd_c4_f32": (c_i, [c_fp, c_i, c_i, c_i, c_fp, c_i, c_i, c_i, c_fp, c_i, c_i, c_i, c_fp, c_fp,
                                          c_i, c_i, c_i, c_i, c_fp]),
    "cnm_packed_winograd_rows_floats": (c_sz, [c_i, c_i, c_i, c_i, c_i]),
    "cnm_pack_winograd_rows_bn_f32": (c_i, [c_fp, c_fp, c_fp, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_fp, c_fp]),
    "cnm_conv_rows_winograd_c4_f32": (c_i, [c_fp, c_i, c_i, c_i, c_fp, c_i, c_i, c_i, c_fp, c_i, c_i, c_i, c_fp, c_fp,
                                            c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_fp]),
    "cnm_conv_rows_winograd_sync_c4_f32": (c_i, [c_fp, c_i, c_i, c_i, c_fp, c_i, c_i, c_i, c_fp, c_i, c_i, c_i, c_fp, c_fp,
                                                 c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_fp, c_sz, c_fp]),
    "cnm_tune_rows7_staged": (c_i, [c_i]),
    "cnm_upsample2x_c4_f32": (c_i, [c_fp, c_i, c_i, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_fp]),
    "cnm_pack_head_f32": (c_i, [c_fp, c_i, c_fp, c_fp]),
    "cnm_head_sigmoid_c4_f32": (c_i, [c_fp, c_i, c_i, c_i, c_fp, c_fp, c_f, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_fp]),
    "cnm_refine_assemble_c4_f32": (c_i, [c_fp, c_fp, c_ll, c_fp, c_i, c_i, c_fp, c_i, c_i, c_fp, c_i, c_i, c_i, c_i, c_fp]),
    "cnm_refine_assemble_multi_c4_f32": (c_i, [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_fp]),
    "cnm_nchw_to_c4_f32": (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_fp]),
    "cnm_c4_to_nchw_f32": (c_i, [c_fp, c_i, c_i, c_fp, c_i, c_i, c_i, c_i, c_fp]),
    "cnm_net_num_layers": (c_i, [c_i]),
    "cnm_net_layer": (c_i, [c_i, c_i, C.POINTER(LayerInfo)]),
    "cnm_depthnet_workspace_floats": (c_sz, [c_i, c_i, c_i, c_i]),
    "cnm_depthnet_forward_f32": (c_i, [C.POINTER(LayerWeights), c_f, c_i, c_fp, c_fp, c_fp, c_fp,
                                       c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_sz, c_i, c_i, c_i, c_i, c_fp]),
    "cnm_refinenet_workspace_floats": (c_sz, [c_i, c_i, c_i]),
    "cnm_refinenet_forward_f32": (c_i, [C.POINTER(LayerWeights), c_f, c_fp, c_fp, c_ll, c_fp, c_i, c_i, c_fp, c_i, c_i,
                                        c_fp, c_fp, c_fp, c_fp, c_sz, c_i, c_i, c_i, c_fp]),
    "cnm_refinenet_forward_multi_f32": (c_i, [C.POINTER(LayerWeights), c_f, c_fp, c_fp, c_i, c_fp, c_fp, c_fp,
                                              c_fp, c_sz, c_i, c_i, c_i, c_fp]),
    "cnm_packed_conv_halfs": (c_sz, [c_i, c_i, c_i]),
    "cnm_pack_conv_bn_f16": (c_i, [c_fp] * 6 + [c_f, c_i, c_i, c_i, c_i, c_fp, c_fp, c_fp]),
    "cnm_conv2d_c8_f16": (c_i, [c_fp, c_i, c_i, c_i, c_fp, c_i, c_i, c_i, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_fp]),
    "cnm_conv2d_cat2_c8_f16": (c_i, [c_fp, c_i, c_i, c_i, c_fp, c_i, c_i, c_i, c_fp, c_i, c_i, c_i, c_fp, c_fp,
                                     c_i, c_i, c_i, c_i, c_i, c_i, c_fp]),
    "cnm_tune_gldsx": (c_i, [c_i]),
    "cnm_tune_sweep_store": (c_i, [c_i, c_fp]),
    "cnm_calibrate_sweep_store_floats": (c_sz, []),
    "cnm_calibrate_sweep_store": (c_i, [c_fp, c_sz, c_fp, c_fp]),
    "cnm_decide_sweep_store": (c_i, [c_i, c_fp]),
    "cnm_debug_sweep_timing_arm": (c_i, [c_i]),
    "cnm_debug_sweep_timing_read": (c_i, [c_fp, c_i]),
    "cnm_planesweep_cat_c8_f16": (c_i, [c_fp, c_fp, c_fp, c_fp, c_fp, c_sz, c_i, c_i, c_i, c_i, c_i, c_d, c_d, c_fp]),
    "cnm_upsample2x_c8_f16": (c_i, [c_fp, c_i, c_i, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_fp]),
    "cnm_head_sigmoid_c8_f16": (c_i, [c_fp, c_i, c_i, c_i, c_fp, c_fp, c_f, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_fp]),
    "cnm_refine_assemble_multi_c8_f16": (c_i, [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_fp]),
    "cnm_nchw_to_c8_f16": (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_fp]),
    "cnm_c8_to_nchw_f16": (c_i, [c_fp, c_i, c_i, c_fp, c_i, c_i, c_i, c_i, c_fp]),
    "cnm_depthnet_workspace_floats_f16": (c_sz, [c_i, c_i, c_i, c_i]),
    "cnm_depthnet_forward_f16": (c_i, [C.POINTER(LayerWeights), c_f, c_i, c_fp, c_fp, c_fp, c_fp,
                                       c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_sz, c_i, c_i, c_i, c_i, c_fp]),
    "cnm_depthnet_forward_strided_f32": (c_i, [C.POINTER(LayerWeights), c_f, c_i, c_fp, c_ll, c_fp, c_ll, c_fp, c_ll, c_fp, c_ll,
                                               c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_sz, c_i, c_i, c_i, c_i, c_fp]),
    "cnm_depthnet_forward_strided_f16": (c_i, [C.POINTER(LayerWeights), c_f, c_i, c_fp, c_ll, c_fp, c_ll, c_fp, c_ll, c_fp, c_ll,
                                               c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_sz, c_i, c_i, c_i, c_i, c_fp]),
    "cnm_homography_terms_strided_f32": (c_i, [c_fp, c_ll, c_fp, c_ll, c_fp, c_i, c_i, c_fp]),
    "cnm_planesweep_cat_strided_c4_f32": (c_i, [c_fp, c_ll, c_fp, c_ll, c_fp, c_fp, c_fp, c_sz, c_i, c_i, c_i, c_i, c_i, c_d, c_d, c_fp]),
    "cnm_planesweep_cat_strided_c8_f16": (c_i, [c_fp, c_ll, c_fp, c_ll, c_fp, c_fp, c_fp, c_sz, c_i, c_i, c_i, c_i, c_i, c_d, c_d, c_fp]),
    "cnm_refinenet_forward_multi_f16": (c_i, [C.POINTER(LayerWeights), c_f, c_fp, c_fp, c_i, c_fp, c_fp, c_fp,
                                              c_fp, c_sz, c_i, c_i, c_i, c_fp]),
    "cnm_packed_dgrad_floats": (c_sz, [c_i, c_i, c_i]),
    "cnm_pack_conv_dgrad_f32": (c_i, [c_fp, c_i, c_i, c_i, c_i, c_fp, c_fp]),
    "cnm_conv2d_dgrad_c4_f32": (c_i, [c_fp, c_i, c_i, c_i, c_fp, c_i, c_i, c_i, c_fp, c_i, c_i, c_i, c_i, c_i, c_fp]),
    "cnm_conv2d_wgrad_workspace_floats": (c_sz, [c_i, c_i, c_i, c_i, c_i, c_i]),
    "cnm_conv3x3_wgrad_winograd_workspace_floats": (c_sz, [c_i, c_i, c_i, c_i, c_i]),
    "cnm_conv7x7_wgrad_winograd_workspace_floats": (c_sz, [c_i, c_i, c_i, c_i, c_i]),
    "cnm_conv7x7_wgrad_winograd_c4_f32": (c_i, [c_fp, c_i, c_i, c_i, c_fp, c_i, c_i, c_i, c_fp, c_fp, c_sz, c_i, c_i, c_i, c_i, c_fp]),
    "cnm_conv5x5_wgrad_winograd_workspace_floats": (c_sz, [c_i, c_i, c_i, c_i, c_i]),
    "cnm_conv5x5_wgrad_winograd_c4_f32": (c_i, [c_fp, c_i, c_i, c_i, c_fp, c_i, c_i, c_i, c_fp, c_fp, c_sz, c_i, c_i, c_i, c_i, c_fp]),
    "cnm_conv_s2_wgrad_winograd_workspace_floats": (c_sz, [c_i, c_i, c_i, c_i, c_i, c_i]),
    "cnm_conv_s2_wgrad_winograd_c4_f32": (c_i, [c_fp, c_i, c_i, c_i, c_fp, c_i, c_i, c_i, c_fp, c_fp, c_sz, c_i, c_i, c_i, c_i, c_i, c_fp]),
    "cnm_conv3x3_wgrad_winograd_c4_f32": (c_i, [c_fp, c_i, c_i, c_i, c_fp, c_i, c_i, c_i, c_fp, c_fp, c_sz, c_i, c_i, c_i, c_i, c_fp]),
    "cnm_conv2d_wgrad_c4_f32": (c_i, [c_fp, c_i, c_i, c_i, c_fp, c_i, c_i, c_i, c_fp, c_fp, c_sz, c_i, c_i, c_i, c_i, c_i, c_i, c_fp]),
    "cnm_bn_train_forward_c4_f32": (c_i, [c_fp, c_fp, c_fp, c_fp, c_fp, c_f, c_f, c_i, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_fp]),
    "cnm_bn_train_backward_c4_f32": (c_i, [c_fp] * 6 + [c_i, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_fp]),
    "cnm_bn_train_forward_z_c4_f32": (c_i, [c_fp, c_fp, c_fp, c_fp, c_fp, c_f, c_f, c_i, c_fp, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_fp]),
    "cnm_bn_train_backward_z_c4_f32": (c_i, [c_fp] * 6 + [c_i, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_fp]),
    "cnm_bn_train_forward_zg_c4_f32": (c_i, [c_fp, c_fp, c_fp, c_fp, c_fp, c_f, c_f, c_i, c_fp, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_fp]),
    "cnm_bn_train_backward_zg_c4_f32": (c_i, [c_fp] * 6 + [c_i, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_fp]),
    "cnm_bn_train_partials_doubles": (c_sz, [c_i, c_i, c_i, c_i, c_i]),
    "cnm_bn_train_forward_p_c4_f32": (c_i, [c_fp, c_fp, c_fp, c_fp, c_fp, c_f, c_f, c_i, c_fp, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_fp]),
    "cnm_bn_train_backward_p_c4_f32": (c_i, [c_fp] * 7 + [c_i, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_fp]),
    "cnm_bn_train_backward_zgb_c4_f32": (c_i, [c_fp] * 6 + [c_i, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_fp]),
    "cnm_head_backward_workspace_doubles": (c_sz, [c_i]),
    "cnm_head_backward_c4_f32": (c_i, [c_fp, c_i, c_i, c_i, c_fp, c_fp, c_fp, c_f, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_fp]),
    "cnm_masked_l1_workspace_doubles": (c_sz, []),
    "cnm_masked_l1_f32": (c_i, [c_fp, c_fp, c_fp, c_ll, c_fp, c_fp, c_fp]),
    "cnm_masked_l1_backward_f32": (c_i, [c_fp, c_fp, c_fp, c_fp, c_fp, c_ll, c_fp, c_fp, c_fp]),
    "cnm_normal_cos_workspace_doubles": (c_sz, [c_i]),
    "cnm_normal_cos_terms_f32": (c_i, [c_fp, c_fp, c_fp, c_i, c_i, c_fp, c_fp, c_fp, c_fp]),
    "cnm_normal_cos_terms_backward_f32": (c_i, [c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_fp, c_fp]),
    "cnm_upsample2x_backward_c4_f32": (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_i, c_fp]),
    "cnm_depth2normal_f32": (c_i, [c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_fp]),
    "cnm_plane_normals_f32": (c_i, [c_fp, c_fp, c_fp, c_i, c_fp, c_i, c_i, c_i, c_i, c_fp]),
    "cnm_depth2normal_backward_f32": (c_i, [c_fp] * 6 + [c_i, c_i, c_i, c_i, c_i, c_fp]),
    "cnm_inverse_warp_backward_depth_f32": (c_i, [c_fp] * 7 + [c_i, c_i, c_i, c_i, c_fp]),
    "cnm_intrinsics_inverse_f32": (c_i, [c_fp, c_ll, c_fp, c_i, c_fp]),
    "cnm_inverse_warp_f32": (c_i, [c_fp] * 6 + [c_i, c_i, c_i, c_i, c_fp]),
    "cnm_inverse_warp_pad_f32": (c_i, [c_fp] * 6 + [c_i, c_i, c_i, c_i, c_i, c_fp]),
    # host twins (csrc/host_twins.cpp, EngHost in csrc/nets.hip): HOST pointers, no stream
    "cnm_homography_terms_cpu": (c_i, [c_fp, c_fp, c_fp, c_i, c_i]),
    "cnm_planesweep_volume_nchw_cpu": (c_i, [c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_d, c_d]),
    "cnm_planesweep_cat_c4_cpu": (c_i, [c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_d, c_d]),
    "cnm_packed_conv_floats_cpu": (c_sz, [c_i, c_i, c_i]),
    "cnm_pack_conv_bn_cpu": (c_i, [c_fp] * 6 + [c_f, c_i, c_i, c_i, c_i, c_fp, c_fp]),
    "cnm_pack_head_cpu": (c_i, [c_fp, c_i, c_fp]),
    "cnm_conv2d_cat2_c4_cpu": (c_i, [c_fp, c_i, c_i, c_i, c_fp, c_i, c_i, c_i, c_fp, c_i, c_i, c_i, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i]),
    "cnm_upsample2x_c4_cpu": (c_i, [c_fp, c_i, c_i, c_fp, c_i, c_i, c_i, c_i, c_i, c_i]),
    "cnm_head_sigmoid_c4_cpu": (c_i, [c_fp, c_i, c_i, c_i, c_fp, c_fp, c_f, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i]),
    "cnm_nchw_to_c4_cpu": (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_i]),
    "cnm_c4_to_nchw_cpu": (c_i, [c_fp, c_i, c_i, c_fp, c_i, c_i, c_i, c_i]),
    "cnm_intrinsics_inverse_cpu": (c_i, [c_fp, c_ll, c_fp, c_i]),
    "cnm_depth2normal_cpu": (c_i, [c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i]),
    "cnm_inverse_warp_cpu": (c_i, [c_fp] * 6 + [c_i, c_i, c_i, c_i]),
    "cnm_depthnet_workspace_floats_cpu": (c_sz, [c_i, c_i, c_i, c_i]),
    "cnm_depthnet_forward_cpu": (c_i, [C.POINTER(LayerWeights), c_f, c_i, c_fp, c_fp, c_fp, c_fp,
                                       c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_sz, c_i, c_i, c_i, c_i]),
    "cnm_refinenet_workspace_floats_cpu": (c_sz, [c_i, c_i, c_i]),
    "cnm_refinenet_forward_cpu": (c_i, [C.POINTER(LayerWeights), c_f, c_fp, c_fp, c_ll, c_fp, c_i, c_i, c_fp, c_i, c_i,
                                        c_fp, c_fp, c_fp, c_fp, c_sz, c_i, c_i, c_i]),
    "cnm_refinenet_forward_multi_cpu": (c_i, [C.POINTER(LayerWeights), c_f, c_fp, c_fp, c_i, c_fp, c_fp, c_fp,
                                              c_fp, c_sz, c_i, c_i, c_i]),
}

NET_DEPTH, NET_REFINE = 0, 1


class EngineError(RuntimeError):
    pass


_lib = None


def load():
    """dlopen the engine (after torch, so that both share ONE libamdhip64.so.7)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EngineError(
            "libcnm_engine.so not found at %s -- build it with `python -m cnmnet_amd.build` "
            "(there is no CPU fallback in this package)" % LIB_PATH)
    import torch  # noqa: F401  (loads torch's HIP runtime first; same SONAME as the one we link)
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)          # AttributeError here = header/library mismatch
        fn.restype, fn.argtypes = res, args
    if lib.cnm_abi_version() != 6:
        raise EngineError("libcnm_engine.so ABI version %d, expected 6" % lib.cnm_abi_version())
    # A/B switches without code changes: CNM_TUNE="wino36_staged=2,refine_side_stream=0" calls cnm_tune_<name>(<value>)
    for item in filter(None, os.environ.get("CNM_TUNE", "").split(",")):
        name, _, val = item.partition("=")
        fn = getattr(lib, "cnm_tune_" + name.strip())
        fn(int(val), *([None] * (len(fn.argtypes or (None,)) - 1)))               # knobs with an output pointer (sweep_store) take NULL
    _lib = lib
    return lib


def check(status):
    if status != 0:
        raise EngineError("cnm_engine: %s (status %d)" % (load().cnm_status_string(status).decode(), status))


def net_layers(net):
    lib = load()
    out = []
    for i in range(lib.cnm_net_num_layers(net)):
        info = LayerInfo()
        check(lib.cnm_net_layer(net, i, C.byref(info)))
        out.append(dict(conv_key=info.conv_key.decode(), bn_key=info.bn_key.decode() if info.bn_key else None,
                        Cin=info.Cin, Cout=info.Cout, ksize=info.ksize, stride=info.stride, rot=info.rot,
                        is_head=bool(info.is_head)))
    return out
