"""Generate tests/golden/scannet_loader.npz by RUNNING the imported reference's ScanNet loader
(scannet/dataloader_batch.py ScannetDataset + Resizer + ToTensor, scannet/preprocess.py load_cam / write_cam /
normalize_image) on a small synthetic scene.  Authoring container only.

cv2 is absent here, so the reference runs under a three-function shim: imread = PIL decode (BGR order like cv2),
cvtColor = channel swap, resize = this build's restatement of INTER_LINEAR / INTER_NEAREST.  The fixture therefore pins
file layout, view order, normalisation, depth clipping, NaN handling, camera parsing and scaling, tensor layout and the
plane helpers -- everything except the resize arithmetic itself.  The decoded pixels are stored so the test can rebuild
the scene with lossless files."""
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import import_reference as ir          # noqa: E402
from cnmnet_amd import scannet as sn               # noqa: E402
from cnmnet_amd.eval7scenes import resize_linear   # noqa: E402


def install_cv2_shim():
    from PIL import Image
    cv2 = sys.modules["cv2"]
    cv2.COLOR_BGR2RGB, cv2.INTER_LINEAR, cv2.INTER_NEAREST = 4, 1, 0
    def imread(path, flags=-1):
        a = np.asarray(Image.open(path))
        return a[..., ::-1].copy() if a.ndim == 3 else a.copy()
    def cvtColor(img, code):
        return img[..., ::-1].copy()
    def resize(img, dsize, interpolation=1):
        w, h = dsize
        return resize_linear(img, h, w) if interpolation == 1 else sn.resize_nearest(img, h, w)
    cv2.imread, cv2.cvtColor, cv2.resize = imread, cvtColor, resize
    sys.modules["tensorflow.python.lib.io"].file_io = sys.modules["tensorflow.python.lib.io.file_io"]


def main():
    ir.load()
    install_cv2_shim()
    import importlib
    ref = importlib.import_module("scannet.dataloader_batch")
    pre = importlib.import_module("scannet.preprocess")
    from PIL import Image
    out = {}
    with tempfile.TemporaryDirectory() as root:
        lst = sn.write_synthetic_scene(root, num_frames=5, interval=10, height=48, width=64, seed=5, planes=True)
        scene = os.path.join(root, "scene0000_00")
        ids = [str(i * 10) for i in range(5)]
        out["ids"] = np.array([int(i) for i in ids])
        out["rgb_u8"] = np.stack([np.asarray(Image.open(os.path.join(scene, "rgb", i + ".jpg")).convert("RGB")) for i in ids])
        out["depth_u16"] = np.stack([np.asarray(Image.open(os.path.join(scene, "depth", i + ".png"))) for i in ids])
        out["normal_npy"] = np.stack([np.load(os.path.join(scene, "lg_normal", i + ".npy")) for i in ids])
        out["cam_txt"] = np.array([open(os.path.join(scene, "cameras", i + "_cam.txt")).read() for i in ids])
        out["plane_seg_u8"] = np.asarray(Image.open(os.path.join(scene, "plane_seg", "10.png")))
        out["plane_para"] = np.load(os.path.join(scene, "plane_para", "10.npy"))
        # reference loader: raw sample, then Resizer + ToTensor to a non-integer scale
        ds_raw = ref.ScannetDataset(lst, root, view_num=3, interval=10, depth_scale=2.5)
        raw = ds_raw[1]
        for k in ("rgbs", "depths", "normals", "cameras"):
            out["raw_" + k] = np.asarray(raw[k])
        out["raw_filenames"] = np.array(raw["filenames"])
        ds = ref.ScannetDataset(lst, root, view_num=3, interval=10, depth_scale=2.5,
                                transform=lambda s: ref.ToTensor()(ref.Resizer(96, 64, 40, 32)(s)))
        t = ds[0]
        for k in ("rgbs", "depths", "normals", "cameras"):
            out["t_" + k] = t[k].numpy()
        # 5 views: order of the sources
        ds5 = ref.ScannetDataset(lst, root, view_num=5, interval=10)
        ds5.sample_list = [["scene0000_00", "20"]]
        out["v5_filenames"] = np.array(ds5[0]["filenames"])
        # preprocess.py pieces
        cam = pre.load_cam(open(os.path.join(scene, "cameras", "20_cam.txt")))
        out["load_cam"] = cam
        p = os.path.join(root, "w_cam.txt")
        pre.write_cam(p, cam[0] + 0.125, cam[1][:3, :3] * 1.5)
        out["write_cam_txt"] = np.array(open(p).read())
        out["scale_camera"] = pre.scale_camera(cam, 0.75, 1.25)
        # plane helpers (methods that do not touch self)
        seg = ds_raw.load_seg(os.path.join(scene, "plane_seg", "10.png"))
        out["load_seg"] = seg.copy()
        para, seg2 = ds_raw.process_by_seg(out["plane_para"].copy(), seg.copy(), "s", "10")
        out["pbs_para"], out["pbs_seg"] = para, seg2
        para_x = ds_raw.plane_para_coordinate_exchange(para.copy())
        out["para_exchanged"] = para_x
        out["instance"] = ds_raw.load_plane_instance_seg(seg2, para.shape[0])
        out["normal_from_para"] = ds_raw.normal_from_plane_para(para_x, para.shape[0], seg2)
        rs = ref.Resizer(96, 64, 40, 32)
        out["instance_resized"] = rs.scale_instance_segs(out["instance"][None], 32, 40)
    np.savez_compressed(os.path.join(HERE, "scannet_loader.npz"), **out)
    print("scannet golden:", {k: getattr(v, "shape", None) for k, v in out.items()})


if __name__ == "__main__":
    main()
