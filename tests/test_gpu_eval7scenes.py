"""7-Scenes harness end to end on the GPU (synthetic sequence: the dataset is not in the image): the engine's frames
scored by cnmnet_amd.eval7scenes against the same frames through the CPU oracle."""
import numpy as np
import pytest
import torch

from cnmnet_amd import eval7scenes as e7, synthetic as syn
from conftest import torch_state
from oracle import ref_arrangement as ra

pytestmark = pytest.mark.gpu


def _load(module, seed):
    shapes = {k: tuple(v.shape) for k, v in module.state_dict().items()}
    module.load_state_dict(torch_state(syn.state_dict_like(shapes, seed=seed, randomize_bn=True)))
    return module.eval()


def test_sequence_metrics_match_oracle(tmp_path):
    from cnmnet_amd.depthnet import depthNet, DepthRefineNet
    from cnmnet_amd.pipeline import FramePipeline
    dev = torch.device("cuda:0")
    seq = str(tmp_path / "chess" / "seq-03")
    e7.write_synthetic_sequence(seq, num_frames=32, height=96, width=128, seed=5)
    H, W = 64, 96
    pipe = FramePipeline(_load(depthNet(3.0), 21).to(dev), _load(DepthRefineNet(32, 3.0), 22).to(dev), k_size=9, normals=False)
    errs, agg = e7.evaluate_sequence(pipe, seq, H, W, views=3, batch=3, device=dev)
    assert len(errs) == 4 and all(np.isfinite(v) for v in agg.values())          # reference frames 12, 15, 18, 21 (every 3rd of 10..21, eval.py:408-409)
    # the same samples through the oracle (CPU, reference arrangement)
    files = e7.sequence_files(seq)
    dn, rn = _load(ra.DepthNetCPU(3.0, 64), 21), _load(ra.DepthRefineNetCPU(32, 3.0), 22)
    want = []
    for r, src in e7.sample_indices(len(files), 3)[:2]:
        s = [e7.load_sample(files[i], H, W) for i in [r] + src]
        T = lambda k, i: torch.from_numpy(s[i][k])[None]
        with torch.no_grad():
            o = ra.frame_forward(dn, rn, T(0, 0), T(0, 1), T(0, 2), T(2, 0), T(2, 1), T(2, 2), k_size=9, normals=False)
        want.append(e7.frame_errors(s[0][1], e7.depth_from_idepth(o["disp"].numpy().reshape(H, W))))
    for got, w in zip(errs[:2], want):
        for k in w:
            assert abs(got[k] - w[k]) < 2e-3 * max(1.0, abs(w[k])), (k, got[k], w[k])


def test_checkpoint_loading_conventions():
    from cnmnet_amd.depthnet import depthNet, DepthRefineNet
    a, b = depthNet(3.0), DepthRefineNet(32, 3.0)
    ck = {"depth_network_state_dict": {"module." + k: v + 1 for k, v in a.state_dict().items()},
          "depth_refine_network_state_dict": {k: v + 2 for k, v in b.state_dict().items()}}
    a2, b2 = depthNet(3.0), DepthRefineNet(32, 3.0)
    e7.load_checkpoint(ck, a2, b2)
    k = "conv1.0.weight"
    assert torch.equal(a2.state_dict()[k], a.state_dict()[k] + 1) and torch.equal(b2.state_dict()[k], b.state_dict()[k] + 2)
    a3 = depthNet(3.0)
    e7.load_checkpoint({"state_dict": a.state_dict()}, a3)                        # eval.py:196 fallback key
    assert torch.equal(a3.state_dict()[k], a.state_dict()[k])
