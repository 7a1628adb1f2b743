"""The algebra behind the stride-2 forms of the staged 36-point kernel (cnmnet_amd/csrc/conv_winograd4.hip pack_winograd36_kernel<R, S2>,
conv_winograd4s.hip S2 / phase scatter), checked on the CPU against torch's own stride-2 convolution and its transpose:
  * a stride-2 k x k convolution (k = 5, 7; pad k // 2) is the sum over the four pixel phases of the input of stride-1 convolutions
    with the R x R sub-filters w[2 jy + py - o][2 jx + px - o] (R = 3, o = 0 / R = 4, o = 1), window starting o + 1 phase pixels
    before the output pixel -- the index formula of the pack kernel;
  * Winograd F(3,4) on the six points {0, +-1, +-2, inf}: the matrices used for the 7x7 layer;
  * the data gradient of a stride-2 convolution is four stride-1 convolutions of dY written to the four pixel phases of dX
    (cnmnet_amd.autograd._stride2_dgrad_phases, _taps4).
No GPU, no engine library."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F


@pytest.mark.parametrize("k", [5, 7])
def test_stride2_is_four_phase_convolutions(k):
    rng = np.random.default_rng(k)
    N, Cin, Cout, H, W = 2, 3, 4, 12, 16
    x = torch.from_numpy(rng.standard_normal((N, Cin, H, W)))
    w = torch.from_numpy(rng.standard_normal((Cout, Cin, k, k)))
    want = F.conv2d(x, w, stride=2, padding=k // 2)
    R, o = (3, 0) if k == 5 else (4, 1)
    lead = o + 1                                                         # phase pixels before the output pixel
    got = torch.zeros_like(want)
    for py in range(2):
        for px in range(2):
            sub = torch.zeros(Cout, Cin, R, R, dtype=torch.float64)
            for jy in range(R):
                for jx in range(R):
                    ky, kx = 2 * jy + py - o, 2 * jx + px - o
                    if 0 <= ky < k and 0 <= kx < k:
                        sub[:, :, jy, jx] = w[:, :, ky, kx]
            phase = F.pad(x[:, :, py::2, px::2], (lead, R - 1 - lead, lead, R - 1 - lead))   # zero padding of the phase image
            got += F.conv2d(phase, sub)
    assert got.shape == want.shape and float((got - want).abs().max()) < 1e-12


def test_winograd_f34_matrices():
    BT = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], float)
    G4 = np.array([[1 / 4, 0, 0, 0], [-1 / 6] * 4, [-1 / 6, 1 / 6, -1 / 6, 1 / 6], [1 / 24, 1 / 12, 1 / 6, 1 / 3], [1 / 24, -1 / 12, 1 / 6, -1 / 3], [0, 0, 0, 1]])
    AT = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 1]], float)
    rng = np.random.default_rng(0)
    d, g = rng.standard_normal((6, 6)), rng.standard_normal((4, 4))
    y = AT @ ((G4 @ g @ G4.T) * (BT @ d @ BT.T)) @ AT.T                   # F(3x3,4x4): 36 products for 9 outputs of 16 taps
    want = np.array([[sum(d[i + p, j + q] * g[p, q] for p in range(4) for q in range(4)) for j in range(3)] for i in range(3)])
    assert np.abs(y - want).max() < 1e-12


@pytest.mark.parametrize("k", [3, 5, 7])
def test_stride2_data_gradient_is_a_phase_scatter(k):
    from cnmnet_amd.autograd import _stride2_dgrad_phases, _taps4
    rng = np.random.default_rng(10 + k)
    N, Cin, Cout, Ho, Wo = 2, 3, 4, 6, 7
    w = torch.from_numpy(rng.standard_normal((Cout, Cin, k, k)))
    dy = torch.from_numpy(rng.standard_normal((N, Cout, Ho, Wo)))
    want = F.conv_transpose2d(dy, w, stride=2, padding=k // 2, output_padding=1)
    got = torch.zeros_like(want)
    for a, b, wp in _stride2_dgrad_phases(w):
        K = wp.shape[2]
        got[:, :, a::2, b::2] = F.conv2d(dy, wp, padding=K // 2)
        # the 4-tap form the F(3x3,4x4) launch uses for the 7x7 layer: taps at offsets -1 .. 2
        if k == 7:
            w4 = _taps4(wp)
            assert w4.shape[2:] == (4, 4)
            assert float((F.conv2d(F.pad(dy, (1, 2, 1, 2)), w4) - got[:, :, a::2, b::2]).abs().max()) < 1e-12
    assert float((got - want).abs().max()) < 1e-12


@pytest.mark.parametrize("k", [3, 5, 7])
def test_stride2_phase_filters_by_one_gather(k):
    """autograd._stride2_dgrad_cat (one gather from the weight) = the four filters of _stride2_dgrad_phases concatenated along the
    output channels (as 4x4 taps for k = 7), bit for bit; and the concatenation reproduces conv_transpose2d phase by phase."""
    from cnmnet_amd.autograd import _stride2_dgrad_phases, _stride2_dgrad_cat, _taps4
    rng = np.random.default_rng(20 + k)
    N, Cin, Cout, Ho, Wo = 2, 5, 6, 5, 6
    w = torch.from_numpy(rng.standard_normal((Cout, Cin, k, k)))
    phases = _stride2_dgrad_phases(w)
    ref = torch.cat([_taps4(wp) if k == 7 else wp for _, _, wp in phases], 0)
    got = _stride2_dgrad_cat(w)
    assert got.shape == ref.shape and torch.equal(got.contiguous(), ref)
    dy = torch.from_numpy(rng.standard_normal((N, Cout, Ho, Wo)))
    want = F.conv_transpose2d(dy, w, stride=2, padding=k // 2, output_padding=1)
    T = got.shape[2]
    full = F.conv2d(F.pad(dy, (1, T - 2, 1, T - 2)), got)                 # taps at offsets -1 .. T - 2: [N, 4 * Cin, Ho, Wo]
    for a in (0, 1):
        for b in (0, 1):
            p = 2 * a + b
            assert float((full[:, p * Cin:(p + 1) * Cin] - want[:, :, a::2, b::2]).abs().max()) < 1e-12


def test_split_sources_and_staged_totals():
    """autograd.SplitSources = strided slices with an interleaving backward (unused outputs count as zero gradients);
    losses.total / row_totals / mean_all = sum / per-sample sums / mean at any size, gradients included."""
    from cnmnet_amd.autograd import SplitSources
    from cnmnet_amd.depthnet.losses import total, row_totals, mean_all
    x = torch.randn(6, 2, 3, dtype=torch.float64, requires_grad=True)
    a, b, c = SplitSources.apply(x, 3)
    assert torch.equal(a, x[0::3]) and torch.equal(b, x[1::3]) and torch.equal(c, x[2::3])
    (a.sum() + (c * 2).sum()).backward()
    g = torch.zeros_like(x); g[0::3] = 1; g[2::3] = 2
    assert torch.equal(x.grad, g)
    for shape in ((4, 1, 192, 256), (3, 1, 177, 233), (1, 1, 1, 5), (2, 3, 1025), (7, 1300)):
        y = torch.rand(shape, dtype=torch.float64, requires_grad=True)
        assert abs(float(total(y) - y.sum())) < 1e-9 and abs(float(mean_all(y) - y.mean())) < 1e-12
        assert float((row_totals(y) - y.flatten(1).sum(1)).abs().max()) < 1e-9
        m = y > 0.5
        assert int(total(m)) == int(m.sum())
        total(y * 2).backward()
        assert torch.equal(y.grad, torch.full_like(y, 2.0))
