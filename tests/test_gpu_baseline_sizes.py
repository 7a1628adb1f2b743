"""GPU (-m gpu): the BASELINE.json configurations at their FULL sizes (configs[2..4]) -- too large for whole-output
comparison with the CPU oracle, so: spot checks against the oracle, size-independent properties, memory bounds."""
import numpy as np
import pytest
import torch

from cnmnet_amd import synthetic as syn
from conftest import normal_parity, torch_state
from oracle import closed_form as cf
from oracle import ref_arrangement as ra

pytestmark = pytest.mark.gpu
T = torch.from_numpy


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _load(module, seed, bn=True):
    shapes = {k: tuple(v.shape) for k, v in module.state_dict().items()}
    module.load_state_dict(torch_state(syn.state_dict_like(shapes, seed=seed, randomize_bn=bn)))
    return module


def test_config4_batch4_four_sources(dev):
    """configs[3]: 640x480, 96 planes, 1 ref + 4 src, batch 4 -> 16 (ref, src) pairs in one call (reference
    eval.py:635-663).  (a) one 64x64 crop of the plane-sweep volume of EVERY pair against the float64 closed form;
    (b) the frame pipeline at that size: finite, inverse depth inside (0, idepth_scale), unit normals, peak memory."""
    from cnmnet_amd import ops
    from cnmnet_amd.depthnet import depthNet, DepthRefineNet
    from cnmnet_amd.pipeline import FramePipeline
    B, S, H, W, D = 4, 4, 480, 640, 96
    img, cams = syn.frames(B, S, H, W, seed=17)
    ti, tc = T(img).to(dev), T(cams).to(dev)
    hmkt = ops.homography_terms(tc[:, 0], tc[:, 1:])
    x = ops.plane_sweep_cat_c4(ti[:, 0].contiguous(), ti[:, 1:].contiguous(), hmkt, 3.0, D)
    vol = x[:, :D // 4].permute(0, 1, 4, 2, 3).reshape(B * S, D, H, W)
    rng = np.random.default_rng(5)
    worst = 0.0
    for p in range(B * S):
        b, s = divmod(p, S)
        y0, x0 = int(rng.integers(0, H - 64)), int(rng.integers(0, W - 64))
        want = cf.plane_sweep_volume(img[b:b + 1, 0], img[b:b + 1, 1 + s], cams[b:b + 1, 0], cams[b:b + 1, 1 + s], 3.0, D,
                                     window=(y0, y0 + 64, x0, x0 + 64))[0]
        got = vol[p, :, y0:y0 + 64, x0:x0 + 64].cpu().numpy()
        err = np.abs(got - want)
        worst = max(worst, float(err.max()))
        assert err.max() < 2e-3 and np.median(err) < 3e-5, (p, float(err.max()), float(np.median(err)))
    # the reference channels of the concatenated conv input (depthNet_model.py:233)
    assert torch.equal(x[:, D // 4, :, :, :3].permute(0, 3, 1, 2), ti[:, 0].repeat_interleave(S, 0))
    del x, vol
    torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
    pipe = FramePipeline(_load(depthNet(3.0, D), 3).to(dev).eval(), _load(DepthRefineNet(32, 3.0), 4).to(dev).eval(), k_size=9)
    out = pipe(ti, tc)
    torch.cuda.synchronize()
    peak = torch.cuda.max_memory_allocated() / 2**30
    print("config 4 (B=4, S=4, 640x480, D=96): plane-sweep crops worst |err| %.1e; peak memory %.1f GiB" % (worst, peak))
    assert peak < 48.0, peak                                             # 288 GB HBM; the workspace is sized per pair, not aliased
    assert torch.isfinite(out["disp"]).all() and torch.isfinite(out["normal"]).all() and torch.isfinite(out["prob"]).all()
    assert float(out["disp"].min()) > 0 and float(out["disp"].max()) < 3.0
    n = out["normal"].norm(dim=1)
    assert bool((((n - 1).abs() < 1e-4) | (n < 1e-6)).all())
    # duplicating the sources of one frame reproduces the 2-source result of that frame (the fusion averages equals)
    a = pipe(torch.cat((ti[:1, :3], ti[:1, 1:3]), 1), torch.cat((tc[:1, :3], tc[:1, 1:3]), 1))
    b2 = pipe(ti[:1, :3].contiguous(), tc[:1, :3].contiguous())
    assert float((a["disp"] - b2["disp"]).abs().max()) < 2e-4


def test_config4_one_frame_vs_oracle(dev):
    """configs[3] at its full size against the CPU oracle: one frame of 1 ref + 4 src at 640x480 with
    96 planes through oracle.ref_arrangement (DepthNetCPU(3.0, 96) on the four pairs, even / odd averaging, the refine net,
    Depth2normal k = 9 -- about a minute of host time) and through the engine as ONE batch entry of a B = 2 call (so the 4 x 4
    and 2 x 8 tile blocks, the 120x160 ... 15x20 feature maps and their stream-K ranges are the ones a real config-4 call
    picks).  Inverse depth / probability max < 1e-3; normals as in test_bench_configuration_vs_oracle."""
    from cnmnet_amd.depthnet import depthNet, DepthRefineNet
    from cnmnet_amd.pipeline import FramePipeline
    B, S, H, W, D = 2, 4, 480, 640, 96
    img, cams = syn.frames(B, S, H, W, seed=23)

    def load(module, seed, head_scale):                                  # heads out of sigmoid saturation (test_bench_configuration_vs_oracle)
        shapes = {k: tuple(v.shape) for k, v in module.state_dict().items()}
        w = syn.state_dict_like(shapes, seed=seed, randomize_bn=True)
        w = {k: (v * head_scale if (v.ndim == 4 and v.shape[0] == 1) else v) for k, v in w.items()}
        module.load_state_dict(torch_state(w))
        return module.eval()

    pipe = FramePipeline(load(depthNet(3.0, D), 61, 0.2).to(dev), load(DepthRefineNet(32, 3.0), 62, 0.05).to(dev), k_size=9)
    with torch.no_grad():
        out = pipe(T(img).to(dev), T(cams).to(dev))
    b = 1
    cd, cr = load(ra.DepthNetCPU(3.0, D), 61, 0.2), load(ra.DepthRefineNetCPU(32, 3.0), 62, 0.05)
    with torch.no_grad():
        disp, prob = ra.frame_forward_multi(cd, cr, T(img[b]), T(cams[b]))
        K_inv = T(cams[b:b + 1, 0])[:, 1, :3, :3].inverse()
        normal, _ = ra.depth_to_normal(1.0 / disp.squeeze(1), K_inv, 9)                  # eval.py:452-455
    e_d = float((out["disp"][b:b + 1].cpu() - disp).abs().max())
    e_p = float((out["prob"][b:b + 1].cpu() - prob).abs().max())
    e_fit, e_ref, excluded, q99 = normal_parity(out["normal"][b:b + 1].cpu(), normal, T(cams[b:b + 1, 0]), disp)
    print("config 4 frame vs oracle (640x480, D=96, S=4): inverse depth max %.1e, probability max %.1e, normals q99 %.1e; vs float64 fit max %.1e, "
          "vs reference max %.1e where it is within 5e-4 of the fit (%.2f %% excluded); oracle inverse depth %.2f .. %.2f"
          % (e_d, e_p, q99, e_fit, e_ref, 100 * excluded, float(disp.min()), float(disp.max())))
    assert float(disp.min()) > 0.05 and float(disp.max()) < 2.95            # the comparison sees every pixel (no saturated sigmoid)
    assert e_d < 1e-3 and e_p < 1e-3, (e_d, e_p)
    assert q99 < 1e-3 and e_fit < 1e-3 and e_ref < 1e-3 and excluded < 0.10, (q99, e_fit, e_ref, excluded)


def test_config3_shard_train_step_first_loss(dev):
    """configs[2], one GPU's shard: `train` step (occlusion fusion + Depth2normal k=9) at 256x192, 64 planes, 4 samples
    per GPU (reference train.py:164-310).  The first-step loss against the CPU oracle in train mode on the same batch."""
    from cnmnet_amd.depthnet import depthNet, DepthRefineNet
    from cnmnet_amd.trainer import TrainStep, synthetic_training_sample
    B, H, W = 4, 192, 256
    sample = synthetic_training_sample(B, H, W, seed=31, device="cpu")
    dn, rn = _load(depthNet(3.0), 81).to(dev), _load(DepthRefineNet(32, 3.0), 82).to(dev)
    step = TrainStep(dn, rn, k_size=9)
    logs = step(**{k: v.to(dev) for k, v in sample.items()})
    assert np.isfinite(logs["loss"])
    # oracle: the same loss on the CPU -- the oracle's nets in train mode, its Depth2normal / inverse_warp restatements,
    # torch's inverse for the intrinsics (forward only: ~1 min of CPU)
    cd, cr = _load(ra.DepthNetCPU(3.0), 81).train(), _load(ra.DepthRefineNetCPU(32, 3.0), 82).train()
    ostep = TrainStep(cd, cr, k_size=9, depth2normal=lambda d, k_inv: ra.depth_to_normal(d, k_inv, 9), inverse_warp=ra.inverse_warp,
                      intrinsics_inverse=lambda cam: torch.linalg.inv(cam[:, 1, :3, :3]))
    with torch.no_grad():
        want = float(ostep.losses(**sample)[0])
    print("config 3 shard (B=4, 256x192, k=9): first-step loss engine %.6f, oracle %.6f" % (logs["loss"], want))
    assert abs(logs["loss"] - want) < 2e-3 * max(1.0, abs(want)), (logs["loss"], want)


def test_config5_f16_vs_f32_full_size(dev):
    """configs[4]: fp16 path (f16 MFMA convolutions, fp16 cost volume) at 256x192, 64 planes, 8 frames per GPU
    (batch 16 over 2 GPUs) against the fp32 engine on the same batch.  Stated tolerance (measured: 1.0e-2 / 2.4e-2 /
    3.4e-2 max, 3.5e-4 / 1.7e-4 mean): inverse depth within 2e-2 (depthNet) / 5e-2 (refined) max and 1e-3 mean on a
    [0, 3] range; probability within 5e-2 max, 2e-3 mean."""
    from cnmnet_amd.depthnet import depthNet, DepthRefineNet
    from cnmnet_amd.pipeline import FramePipeline
    B, S, H, W = 8, 2, 192, 256
    img, cams = syn.frames(B, S, H, W, seed=23)
    ti, tc = T(img).to(dev), T(cams).to(dev)
    outs = {}
    for prec in ("f32", "f16"):
        pipe = FramePipeline(_load(depthNet(3.0, 64, precision=prec), 7).to(dev).eval(),
                             _load(DepthRefineNet(32, 3.0, precision=prec), 8).to(dev).eval(), k_size=9, normals=False)
        o = pipe(ti, tc)
        outs[prec] = {k: o[k].float().clone() for k in ("disp", "prob", "disp_a", "disp_b")}
        del pipe
        torch.cuda.empty_cache()
    d = {k: (outs["f16"][k] - outs["f32"][k]).abs() for k in outs["f32"]}
    print("f16 vs f32 at 192x256, 8 frames: disp pairs max %.2e mean %.2e | refined max %.2e mean %.2e | prob max %.2e" % (
        float(d["disp_a"].max()), float(d["disp_a"].mean()), float(d["disp"].max()), float(d["disp"].mean()), float(d["prob"].max())))
    assert float(d["disp_a"].max()) < 2e-2 and float(d["disp_b"].max()) < 2e-2 and float(d["disp_a"].mean()) < 1e-3
    assert float(d["disp"].max()) < 5e-2 and float(d["disp"].mean()) < 1e-3
    assert float(d["prob"].max()) < 5e-2 and float(d["prob"].mean()) < 2e-3


def test_forward_pairs_batch_split_matches_unsplit(dev):
    """depthNet.forward_pairs splits the batch when one call would pass the engine's 32-bit byte offsets
    (depthNet_model.py forward_pairs); forcing the split at a small size must reproduce the unsplit result.  (Not
    bit for bit: the executors choose F(4x4,3x3) or F(2x2,3x3) per layer by tile count, which the pair count changes.)"""
    from cnmnet_amd.depthnet import depthNet
    B, S, H, W = 4, 2, 64, 96
    img, cams = syn.frames(B, S, H, W, seed=3)
    ti, tc = T(img).to(dev), T(cams).to(dev)
    net = _load(depthNet(3.0), 9).to(dev).eval()
    args = (ti[:, 0].contiguous(), ti[:, 1:].contiguous(), tc[:, 0].contiguous(), tc[:, 1:].contiguous())
    with torch.no_grad():
        whole = net.forward_pairs(*args)
        net.max_call_bytes = 512 * H * W * (S + 1)                        # room for one frame (S pairs) per call: four calls
        split = net.forward_pairs(*args)
    assert split[0][0].shape == whole[0][0].shape and split[1].shape == whole[1].shape
    for a, b in zip(whole[0], split[0]):
        assert float((a - b).abs().max()) < 1e-4                          # kernel choice per layer differs with the pair count (fp32 rounding on a [0,3] range)
    assert float((whole[1] - split[1]).abs().max()) < 1e-4 * float(whole[1].abs().max())


def test_pipeline_decides_the_sweep_store_policy_in_its_own_step(dev):
    """[r6] FramePipeline's first call measures the plane sweep's two store policies INSIDE its own step (ops.calibrate_sweep_store_in_step)
    and records the decision for the device before anything can be captured or timed; later calls and a HIP-graph capture of the
    pipeline neither sample nor change it, and the outputs do not depend on it."""
    import ctypes
    from cnmnet_amd import _lib, ops
    from cnmnet_amd.pipeline import FramePipeline, GraphedFramePipeline
    from cnmnet_amd.depthnet import depthNet, DepthRefineNet
    lib = _lib.load()
    B, S, H, W, D = 4, 2, 192, 256, 64
    img, cams = syn.frames(B, S, H, W, seed=5)
    img, cams = T(img).to(dev), T(cams).to(dev)
    idx = torch.device(dev).index if torch.device(dev).index is not None else torch.cuda.current_device()
    lib.cnm_tune_sweep_store(-1, None)
    ops._SWEEP_STORE_IN_STEP.discard(idx)
    try:
        pipe = FramePipeline(_load(depthNet(3.0, D), 3).to(dev).eval(), _load(DepthRefineNet(32, 3.0), 4).to(dev).eval(), k_size=9)
        out = pipe(img, cams)
        med = (ctypes.c_float * 2)()
        pol = lib.cnm_tune_sweep_store(99, ctypes.cast(med, ctypes.c_void_p))
        assert pol in (0, 2) and 5.0 < med[0] < 500.0 and 5.0 < med[1] < 500.0 and (med[0] < med[1]) == (pol == 0), (pol, med[0], med[1])
        ref = {k: v.clone() for k, v in out.items() if torch.is_tensor(v)}
        g = GraphedFramePipeline(pipe, img, cams)
        got = g()
        torch.cuda.synchronize()
        assert lib.cnm_tune_sweep_store(99, None) == pol
        assert all(torch.equal(got[k], ref[k]) for k in ("disp", "prob"))
        lib.cnm_tune_sweep_store(2 - pol, None)                              # the other policy: the same bytes
        other = pipe(img, cams)
        assert all(torch.equal(other[k], ref[k]) for k in ("disp", "prob"))
    finally:
        lib.cnm_tune_sweep_store(-1, None)
        ops._SWEEP_STORE_IN_STEP.discard(idx)
        ops._SWEEP_STORE_CALIBRATED.discard(idx)
