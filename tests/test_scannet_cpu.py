"""ScanNet loader (cnmnet_amd/scannet.py) against fixtures produced by the imported reference loader
(tests/golden/make_golden_scannet.py), plus the build's extensions.  CPU only."""
import os
import struct
import zlib

import numpy as np
import pytest
import torch

from cnmnet_amd import scannet as sn


@pytest.fixture(scope="module")
def g(golden):
    return golden("scannet_loader.npz")


@pytest.fixture()
def scene(g, tmp_path):
    """The generator's scene rebuilt from the stored decoded pixels with lossless files (rgb as .png)."""
    from PIL import Image
    root = str(tmp_path)
    s = os.path.join(root, "scene0000_00")
    for sub in ("rgb", "depth", "lg_normal", "cameras", "plane_seg", "plane_para"):
        os.makedirs(os.path.join(s, sub))
    for n, i in enumerate(g["ids"]):
        Image.fromarray(g["rgb_u8"][n], "RGB").save(os.path.join(s, "rgb", "%d.png" % i))
        Image.fromarray(g["depth_u16"][n]).save(os.path.join(s, "depth", "%d.png" % i))
        np.save(os.path.join(s, "lg_normal", "%d.npy" % i), g["normal_npy"][n])
        with open(os.path.join(s, "cameras", "%d_cam.txt" % i), "w") as f:
            f.write(str(g["cam_txt"][n]))
        Image.fromarray(g["plane_seg_u8"]).save(os.path.join(s, "plane_seg", "%d.png" % i))
        np.save(os.path.join(s, "plane_para", "%d.npy" % i), g["plane_para"])
    lst = os.path.join(root, "list.txt")
    with open(lst, "w") as f:
        f.write("scene0000_00 10\nscene0000_00 20\nscene0000_00 30\n")
    return root, lst


def test_raw_sample_matches_reference(g, scene):
    root, lst = scene
    ds = sn.ScannetDataset(lst, root, view_num=3, interval=10, depth_scale=2.5, rgb_ext=".png")
    assert len(ds) == 3
    s = ds[1]
    assert list(s["filenames"]) == list(g["raw_filenames"])                 # ref, ref - interval, ref + interval
    for k in ("rgbs", "depths", "normals", "cameras"):
        assert s[k].shape == g["raw_" + k].shape and s[k].dtype == g["raw_" + k].dtype, k
        np.testing.assert_array_equal(s[k], g["raw_" + k], err_msg=k)
    assert (s["depths"] == 0).any() and s["depths"].max() <= 2.5            # the 3 m wall is beyond depth_scale
    assert not np.isnan(s["normals"]).any()


def test_resized_tensors_match_reference(g, scene):
    root, lst = scene
    tf = sn.Compose([sn.Resizer(96, 64, 40, 32), sn.ToTensor()])
    t = sn.ScannetDataset(lst, root, depth_scale=2.5, transform=tf, rgb_ext=".png")[0]
    for k in ("rgbs", "depths", "normals", "cameras"):
        assert tuple(t[k].shape) == g["t_" + k].shape and t[k].dtype == torch.float32, k
        np.testing.assert_array_equal(t[k].numpy(), g["t_" + k], err_msg=k)
    d = t["depths"].numpy().astype(np.float64)
    want = 1.0 / (d + 1e-4); want[(want < 0.02) | (want > 3.0)] = 0
    np.testing.assert_allclose(t["disparities"].numpy(), want, rtol=1e-6)
    assert (t["disparities"].numpy()[d == 0] == 0).all()


def test_five_view_order(g, scene):
    root, lst = scene
    ds = sn.ScannetDataset(lst, root, view_num=5, interval=10, rgb_ext=".png")
    assert ["scene0000_00_" + i for i in ds.view_ids("20")] == list(g["v5_filenames"])


def test_camera_files(g, tmp_path):
    p = str(tmp_path / "c.txt")
    with open(p, "w") as f:
        f.write(str(g["cam_txt"][2]))
    cam = sn.load_cam(p)
    np.testing.assert_array_equal(cam, g["load_cam"])
    with open(p) as f:
        np.testing.assert_array_equal(sn.load_cam(f), g["load_cam"])         # open file objects work like the reference's
    q = str(tmp_path / "w.txt")
    sn.write_cam(q, cam[0] + 0.125, cam[1][:3, :3] * 1.5)
    assert open(q).read() == str(g["write_cam_txt"])
    np.testing.assert_array_equal(sn.scale_camera(cam, 0.75, 1.25), g["scale_camera"])
    assert cam[1, 0, 0] == g["load_cam"][1, 0, 0]                            # input not modified


def test_plane_helpers(g, scene):
    root, _ = scene
    seg = sn.load_seg(os.path.join(root, "scene0000_00", "plane_seg", "10.png"))
    np.testing.assert_array_equal(seg, g["load_seg"])
    para, seg2 = sn.process_by_seg(g["plane_para"].copy(), seg.copy())
    np.testing.assert_array_equal(para, g["pbs_para"]); np.testing.assert_array_equal(seg2, g["pbs_seg"])
    px = sn.plane_para_coordinate_exchange(para.copy())
    np.testing.assert_array_equal(px, g["para_exchanged"])
    inst = sn.load_plane_instance_seg(seg2, para.shape[0])
    np.testing.assert_array_equal(inst, g["instance"])
    np.testing.assert_array_equal(sn.normal_from_plane_para(px, para.shape[0], seg2), g["normal_from_para"])
    rs = np.stack([sn.resize_nearest(m, 32, 40) for m in inst])
    np.testing.assert_array_equal(rs, g["instance_resized"][0])
    with pytest.raises(Exception, match="no planes"):
        sn.process_by_seg(g["plane_para"].copy(), np.full((8, 8), sn.NON_PLANAR, np.uint8))
    tiny = np.full((20, 20), sn.NON_PLANAR, np.uint8); tiny[:3, :3] = 0
    with pytest.raises(Exception, match="wrong plane instance"):
        sn.load_plane_instance_seg(tiny, 1)


def test_plane_and_source_depth_extensions(g, scene):
    root, lst = scene
    tf = sn.Compose([sn.Resizer(64, 48, 64, 48), sn.ToTensor()])
    t = sn.ScannetDataset(lst, root, transform=tf, rgb_ext=".png", source_depths=True, planes=True)[1]
    assert tuple(t["depths"].shape) == (3, 1, 48, 64) and tuple(t["disparities"].shape) == (3, 1, 48, 64)
    assert tuple(t["plane_instance_segs"].shape) == (1, 20, 48, 64) and t["plane_instance_segs"].dtype == torch.uint8
    assert int(t["plane_nums"][0]) == 2 and tuple(t["normals_from_plane_para"].shape) == (1, 3, 48, 64)
    np.testing.assert_array_equal(t["plane_instance_segs"][0].numpy(), g["instance"])
    np.testing.assert_allclose(t["normals_from_plane_para"][0].numpy(), g["normal_from_para"].transpose(2, 0, 1), rtol=1e-6)
    # painted plane normals agree with the depth-derived normal map's axes on both planes
    n = t["normals"][0].numpy(); p = t["normals_from_plane_para"][0].numpy()
    m = t["plane_instance_segs"][0, :2].numpy().astype(bool).any(0)
    assert ((n * p).sum(0)[m] > 0.99).all()


def test_batches_and_rank_shards(scene):
    root, lst = scene
    with open(lst, "a") as f:
        f.write("scene0000_00 20\n")                                          # 4 samples
    dl = sn.training_loader(lst, root, 32, 64, batch_size=2, shuffle=False, rgb_ext=".png", source_depths=True)
    batches = list(dl)
    assert len(batches) == 2
    b = batches[0]
    assert tuple(b["rgbs"].shape) == (2, 3, 3, 32, 64) and tuple(b["cameras"].shape) == (2, 3, 2, 4, 4)
    assert tuple(b["depths"].shape) == (2, 3, 1, 32, 64) and tuple(b["normals"].shape) == (2, 1, 3, 32, 64)
    shards = [sn.ScannetDataset(lst, root, rgb_ext=".png", rank=r, world_size=2).sample_list for r in range(2)]
    assert [len(s) for s in shards] == [2, 2] and shards[0] + shards[1] == sn.ScannetDataset(lst, root).sample_list
    odd = sn.ScannetDataset(lst, root, rank=2, world_size=3).sample_list     # 4 samples over 3 ranks: one each, tail dropped
    assert len(odd) == 1


def test_resize_nearest_and_errors(scene):
    a = np.arange(5 * 7).reshape(5, 7)
    np.testing.assert_array_equal(sn.resize_nearest(a, 5, 7), a)
    np.testing.assert_array_equal(sn.resize_nearest(a, 10, 14), a.repeat(2, 0).repeat(2, 1))
    np.testing.assert_array_equal(sn.resize_nearest(a, 2, 3), a[[0, 2]][:, [0, 2, 4]])    # floor(i * 2.5), floor(j * 7/3)
    root, lst = scene
    ds = sn.ScannetDataset(lst, root, rgb_ext=".jpg")                         # only .png files exist
    with pytest.raises(IOError, match="load image error"):
        ds[0]
    ds = sn.ScannetDataset(lst, root, rgb_ext=".png", depth_scale=0.5)        # nothing closer than 0.8 m
    with pytest.raises(ValueError, match="depth error"):
        ds[0]


def _png16_rgb(path, arr, filter_type=0):
    h, w, _ = arr.shape
    be = arr.astype(">u2").tobytes()
    rows = np.frombuffer(be, np.uint8).reshape(h, w * 6).astype(np.int32)
    out = bytearray()
    for y in range(h):
        line = rows[y]
        if filter_type == 1:                                                 # Sub
            prev = np.concatenate([np.zeros(6, np.int32), line[:-6]]); line = (line - prev) & 255
        elif filter_type == 2:                                               # Up
            prev = rows[y - 1] if y else np.zeros(w * 6, np.int32); line = (line - prev) & 255
        out += bytes([filter_type]) + bytes(line.astype(np.uint8))
    chunk = lambda k, d: struct.pack(">I", len(d)) + k + d + struct.pack(">I", zlib.crc32(k + d) & 0xFFFFFFFF)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 16, 2, 0, 0, 0))
                + chunk(b"IDAT", zlib.compress(bytes(out))) + chunk(b"IEND", b""))


@pytest.mark.parametrize("ft", [0, 1, 2])
def test_png16_normals(scene, ft):
    """Normals stored as 16-bit RGB png decode to (v / 65535 - 0.5) * 2 (dataloader_batch.py:82-88)."""
    root, lst = scene
    rng = np.random.default_rng(3)
    arr = rng.integers(0, 65536, (48, 64, 3)).astype(np.uint16)
    d = os.path.join(root, "scene0000_00", "lg_normal")
    os.remove(os.path.join(d, "10.npy"))
    _png16_rgb(os.path.join(d, "10.png"), arr, ft)
    s = sn.ScannetDataset(lst, root, rgb_ext=".png")[0]
    np.testing.assert_allclose(s["normals"][0], (arr.astype(np.float32) / 65535.0 - 0.5) * 2, rtol=0, atol=1e-7)


def test_synthetic_scene_round_trip(tmp_path):
    lst = sn.write_synthetic_scene(str(tmp_path), num_frames=4, height=32, width=64, planes=True)
    ds = sn.ScannetDataset(lst, str(tmp_path), planes=True, source_depths=True)
    assert len(ds) == 2
    s = ds[0]
    assert s["rgbs"].shape == (3, 32, 64, 3) and s["cameras"].shape == (3, 2, 4, 4)
    assert s["cameras"][1][0][0, 3] > s["cameras"][0][0][0, 3] > s["cameras"][2][0][0, 3]    # previous, reference, next view
