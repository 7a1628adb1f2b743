"""CPU: the C-ABI library loads without a GPU and exports every symbol include/cnm_engine.h
declares; the ctypes table in cnmnet_amd/_lib.py covers exactly that set; host-only entry
points behave.  No GPU compute call is made here (the host twins are exercised in test_host_twins_cpu.py)."""
import ctypes
import os
import re
import subprocess

import pytest

from cnmnet_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "cnm_engine.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return set(re.findall(r"\b(cnm_[a-z0-9_]+)\s*\(", text))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        from cnmnet_amd.build import build
        build(verbose=False)
    return _lib.load()


def test_header_symbols_exported_and_bound(lib):
    declared = _declared()
    assert len(declared) >= 25
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH], text=True)
    exported = set(re.findall(r" T (cnm_[a-z0-9_]+)", out))
    assert declared <= exported, declared - exported
    assert declared == set(_lib.PROTOTYPES), declared ^ set(_lib.PROTOTYPES)


def test_host_entry_points(lib):
    assert lib.cnm_abi_version() == 6
    lo, hi = ctypes.c_double(), ctypes.c_double()
    assert lib.cnm_idepth_range_host(3.0, ctypes.byref(lo), ctypes.byref(hi)) == 0 and (lo.value, hi.value) == (0.1, 3.0)
    assert lib.cnm_idepth_range_host(2.0, ctypes.byref(lo), ctypes.byref(hi)) == 0 and (lo.value, hi.value) == (0.02, 2.0)
    assert lib.cnm_idepth_range_host(2.5, ctypes.byref(lo), ctypes.byref(hi)) == -3     # reference: UnboundLocalError
    assert b"2.0 or 3.0" in lib.cnm_status_string(-3)
    assert lib.cnm_packed_conv_floats(128, 67, 7) == 128 * (((49 * 68 + 15) // 16) * 16)
    assert lib.cnm_depthnet_workspace_floats(1, 200, 256, 64) == 0                      # H not a multiple of 32
    assert lib.cnm_depthnet_workspace_floats(16, 192, 256, 64) * 4 < 3 * 2**30
    # argument validation happens before any launch: null pointers are rejected on a GPU-less host
    assert lib.cnm_depth2normal_f32(0, 0, 0, 0, 1, 8, 8, 9, 0, 0) == -1
    assert lib.cnm_conv2d_c4_f32(0, 1, 0, 1, 0, 16, 0, 64, 0, 0, 1, 8, 8, 3, 1, 1, 0) == -1


def test_layer_tables_match_reference_state_dict():
    """Engine layer tables vs the reference's module structure (via the oracle restatement,
    whose state_dict is verified against the reference in test_oracle_golden.py)."""
    from oracle import ref_arrangement as ra
    for net, mod in ((_lib.NET_DEPTH, ra.DepthNetCPU(3.0)), (_lib.NET_REFINE, ra.DepthRefineNetCPU())):
        sd = mod.state_dict()
        layers = _lib.net_layers(net)
        conv_keys = {k[:-7] for k, v in sd.items() if k.endswith(".weight") and v.dim() == 4}
        assert {L["conv_key"] for L in layers} == conv_keys
        for L in layers:
            w = sd[L["conv_key"] + ".weight"]
            assert tuple(w.shape) == (L["Cout"], L["Cin"], L["ksize"], L["ksize"])
            if L["is_head"]:
                assert L["conv_key"] + ".bias" in sd and L["bn_key"] is None
            else:
                assert L["bn_key"] + ".running_var" in sd and sd[L["bn_key"] + ".weight"].shape[0] == L["Cout"]


def test_product_modules_share_reference_state_dict_and_cpu_behaviour():
    import torch
    from cnmnet_amd.depthnet import depthNet, DepthRefineNet, Depth2normal
    from oracle import ref_arrangement as ra
    for mine, ref in ((depthNet(3.0), ra.DepthNetCPU(3.0)), (DepthRefineNet(32, 3.0), ra.DepthRefineNetCPU(32, 3.0))):
        a, b = mine.state_dict(), ref.state_dict()
        assert list(a) == list(b) and all(a[k].shape == b[k].shape for k in a)
        mine.load_state_dict({"module." + k: v for k, v in b.items()}, strict=False)   # keys only; see eval.py:189-196
        mine.load_state_dict(b)
    # CPU tensors: eval-mode fp32 runs on the library's host twins (tests/test_host_twins_cpu.py); everything else fails loudly --
    # there is no torch fallback and oracle/ is never involved
    net = depthNet(3.0)
    x = torch.zeros(1, 3, 32, 32); cam = torch.eye(4).repeat(1, 2, 1, 1)
    with pytest.raises(_lib.EngineError):
        net.train()(x, x, cam, cam)
    with pytest.raises(_lib.EngineError):
        depthNet(3.0, precision="f16").eval()(x, x, cam, cam)
    with pytest.raises(_lib.EngineError):
        Depth2normal(9)(torch.ones(1, 8, 8, dtype=torch.float64), torch.eye(3, dtype=torch.float64)[None])
    n, p = Depth2normal(9)(torch.ones(1, 8, 8), torch.eye(3)[None])
    assert n.shape == (1, 3, 8, 8) and p.shape == (1, 3, 8, 8)


def test_frame_view_recognises_frame_slices():
    """[r6] depthNet_model._frame_view (host logic, no GPU): slices of a frame tensor whose frames are dense and whole images apart are
    handed to the engine as (pointer, frame stride); anything else is copied."""
    import torch
    from cnmnet_amd.depthnet.depthNet_model import _frame_view
    B, S, H, W = 3, 2, 8, 12
    frames = torch.randn(B, 1 + S, 3, H, W)
    v, st = _frame_view(frames[:, 0], 3 * H * W)
    assert v.data_ptr() == frames.data_ptr() and st == (1 + S) * 3 * H * W
    v, st = _frame_view(frames[:, 1:], 3 * H * W)
    assert v.data_ptr() == frames[:, 1:].data_ptr() and st == (1 + S) * 3 * H * W
    v, st = _frame_view(frames[:1, 0], 3 * H * W)                         # one frame: any stride will do, the dense one is reported
    assert v.data_ptr() == frames.data_ptr() and st == 3 * H * W
    cams = torch.randn(B, 1 + S, 2, 4, 4)
    v, st = _frame_view(cams[:, 1:])
    assert v.data_ptr() == cams[:, 1:].data_ptr() and st == (1 + S) * 32
    wide = torch.randn(B, 4, H, W)
    v, st = _frame_view(wide[:, :3], 3 * H * W)                          # dense frames, but not whole images apart: copied
    assert v.is_contiguous() and v.data_ptr() != wide.data_ptr() and st == 3 * H * W and torch.equal(v, wide[:, :3])
    v, st = _frame_view(frames[:, 0, :, :, ::2], 3 * H * (W // 2))       # not dense inside a frame: copied
    assert v.is_contiguous() and st == 3 * H * (W // 2)
    v, st = _frame_view(frames.flip(0)[:, 0], 3 * H * W)                 # (flip materialises a copy; its slice is an ordinary view again)
    assert st == (1 + S) * 3 * H * W
    d = torch.randn(B, 3, H, W, dtype=torch.float64)
    v, st = _frame_view(d, 3 * H * W)                                    # other dtypes are converted
    assert v.dtype == torch.float32 and st == 3 * H * W
