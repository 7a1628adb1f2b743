"""ScanNet training data for the engine's trainer.

Host-side restatement of the reference's ScanNet loader and its transforms:
  * camera text files `<id>_cam.txt` -- reader `scannet/preprocess.py:29-46`, writer `:130-149`;
  * image normalisation `scannet/preprocess.py:16-26`;
  * `ScannetDataset` (`scannet/dataloader_batch.py:9-81`): list file of "<scene> <image id>" lines, reference view first,
    then the views at `id + interval * (v - view_num // 2)`; depth in millimetres -> metres, < 0.1 m and > depth_scale
    zeroed (`:110-122`); normals from `.npy` (or a 16-bit `.png`), NaN -> 0 (`:82-88, :126-136`);
  * `Resizer` (`:242-336`) and `ToTensor` (`:339-421`);
  * the plane helpers the shipped loader defines but never calls (`:173-239`): `load_seg`, `load_plane_instance_seg`,
    `process_by_seg`, `plane_para_coordinate_exchange`, `normal_from_plane_para`.

Beyond the shipped loader (the reference's train.py reads keys it never produces, SURVEY 0.1 / train.py:145-162):
  * `disparities` -- the formula the reference left commented out (`dataloader_batch.py:115-117, :354-356`):
    1 / (depth + 1e-4), values < 0.02 or > 3.0 zeroed;
  * `source_depths=True` also loads the ground-truth depth of the source views (the warped-depth loss of train.py:287-293
    needs them);
  * `planes=True` reads `<scene>/plane_seg/<id>.png` and `<scene>/plane_para/<id>.npy` and emits `plane_segs`,
    `plane_instance_segs`, `plane_nums`, `normals_from_plane_para` through the helpers above.  The directory names are
    this build's choice (the reference has no call site) -- parity unpinned for the layout, pinned for the helpers.

Files are decoded with PIL (cv2 is absent here).  Resizing restates cv2.resize: INTER_LINEAR's floating-point path for the
already normalised images, INTER_NEAREST's `floor(dst * src / dst_size)` for depth / normals / segmentations.
tests/golden/scannet_loader.npz pins everything except the resize arithmetic itself against the imported reference.
"""
import os

import numpy as np

from .eval7scenes import normalize_image, resize_linear

NON_PLANAR = 20          # dataloader_batch.py:175 ; also the capacity of the instance stack (:186)
MAX_PLANES = 20


# ------------------------------------------------------------------ camera files
def load_cam(file_or_path):
    """[2,4,4] float64: [0] = 4x4 extrinsic, [1][:3,:3] = intrinsics.  Token positions as scannet/preprocess.py:29-46:
    token 0 is the word 'extrinsic', 1..16 the matrix, 17 the word 'intrinsic', 18..26 the 3x3."""
    if isinstance(file_or_path, (str, bytes, os.PathLike)):
        with open(file_or_path, "r") as f:
            words = f.read().split()
    else:
        words = file_or_path.read().split()
    cam = np.zeros((2, 4, 4))
    cam[0] = np.array([float(w) for w in words[1:17]]).reshape(4, 4)
    cam[1, :3, :3] = np.array([float(w) for w in words[18:27]]).reshape(3, 3)
    return cam


def write_cam(path, extrinsic, intrinsic):
    """Inverse of load_cam, same text layout as scannet/preprocess.py:130-149."""
    with open(path, "w+") as f:
        f.write("extrinsic\n")
        for i in range(4):
            f.write(" ".join(str(extrinsic[i][j]) for j in range(4)) + " \n")
        f.write("\nintrinsic\n")
        for i in range(3):
            f.write(" ".join(str(intrinsic[i][j]) for j in range(3)) + " \n")


def scale_camera(cam, scale_x, scale_y):
    """Copy of cam with focal lengths and principal point scaled (preprocess.py:76-87, dataloader_batch.py:300-311)."""
    out = np.array(cam, copy=True)
    out[1, 0, 0] *= scale_x; out[1, 0, 2] *= scale_x
    out[1, 1, 1] *= scale_y; out[1, 1, 2] *= scale_y
    return out


# ------------------------------------------------------------------ resizing
def resize_nearest(img, out_h, out_w):
    """cv2.resize(..., INTER_NEAREST): src index = min(floor(dst * n_in / n_out), n_in - 1)."""
    img = np.asarray(img)
    h, w = img.shape[:2]
    ys = np.minimum(np.floor(np.arange(out_h) * (h / float(out_h))).astype(np.int64), h - 1)
    xs = np.minimum(np.floor(np.arange(out_w) * (w / float(out_w))).astype(np.int64), w - 1)
    return img[ys][:, xs]


# ------------------------------------------------------------------ plane helpers (dataloader_batch.py:173-239)
def load_seg(path):
    """Plane segmentation map; the largest label marks non-planar pixels and becomes 20 (:173-177)."""
    from PIL import Image
    seg = np.array(Image.open(path))
    seg[seg == np.max(seg)] = NON_PLANAR
    return seg


def load_plane_instance_seg(seg, plane_num):
    """[20,H,W] uint8 one-hot masks of planes 0..plane_num-1; a plane under 100 pixels is an error (:179-193)."""
    instance = np.zeros((MAX_PLANES,) + seg.shape, np.uint8)
    for i in range(plane_num):
        instance[i] = seg == i
        if np.sum(seg == i) < 100:
            raise Exception("wrong plane instance")
    return instance


def process_by_seg(plane_para, plane_seg, scene_id=None, image_id=None):
    """Keep the parameters of the planes present in the map and renumber the map 0..n-1 in label order (:195-217).
    Renumbers `plane_seg` IN PLACE like the reference (so a new index can collide with a not yet visited label --
    reproduced as-is)."""
    if len(np.unique(plane_seg)) == 1:
        raise Exception("there are no planes in plane_seg", scene_id, image_id)
    kept, i = [], 0
    for label in np.unique(plane_seg):
        if label == NON_PLANAR:
            continue
        kept.append(plane_para[label])
        plane_seg[plane_seg == label] = i
        i += 1
    return np.stack(kept, axis=0), plane_seg


def plane_para_coordinate_exchange(plane_para):
    """PlaneRCNN's axes -> the depth-derived normal map's: (x, y, z) -> (x, -z, y), in place (:219-231)."""
    tmp = plane_para[:, 1].copy()
    plane_para[:, 1] = -plane_para[:, 2]
    plane_para[:, 2] = tmp
    return plane_para


def normal_from_plane_para(plane_para, plane_num, seg):
    """[H,W,3] float64 unit normals painted per plane, zero elsewhere (:233-241)."""
    normal = np.zeros(seg.shape + (3,))
    for i in range(plane_num):
        normal[seg == i] = plane_para[i]
    normal /= (np.linalg.norm(normal, ord=2, axis=2, keepdims=True) + 1e-5)
    return normal


def disparity_from_depth(depth):
    """dataloader_batch.py:115-117 (commented out there; train.py:162 reads the key)."""
    d = np.reciprocal(np.asarray(depth, np.float64) + 1e-4)
    d[d < 0.02] = 0
    d[d > 3.0] = 0
    return d


# ------------------------------------------------------------------ dataset
class ScannetDataset:
    """Map-style dataset (len / getitem), usable with torch.utils.data.DataLoader.  Arguments as the reference's
    (`dataloader_batch.py:10`); the keyword-only ones are this build's extensions (module docstring)."""

    def __init__(self, list_filepath, root_dir, view_num=3, interval=10, depth_scale=5.0, transform=None, *,
                 source_depths=False, planes=False, rgb_ext=".jpg", rank=0, world_size=1):
        self.list_filepath, self.root_dir = list_filepath, root_dir
        self.view_num, self.interval, self.depth_scale, self.transform = view_num, interval, depth_scale, transform
        self.source_depths, self.planes, self.rgb_ext = source_depths, planes, rgb_ext
        with open(list_filepath, "r") as f:
            self.sample_list = [line.split() for line in f.readlines() if line.strip()]
        if world_size > 1:                                  # contiguous shard per rank, equal lengths (tail dropped)
            from .sharding import shard_range
            lo, hi = shard_range(len(self.sample_list) - len(self.sample_list) % world_size, rank, world_size)
            self.sample_list = self.sample_list[lo:hi]

    def __len__(self):
        return len(self.sample_list)

    def _path(self, scene_id, sub, name):
        return os.path.join(self.root_dir, scene_id, sub, name)

    def _rgb(self, scene_id, image_id):
        from PIL import Image
        path = self._path(scene_id, "rgb", image_id + self.rgb_ext)
        try:
            return normalize_image(np.asarray(Image.open(path).convert("RGB")))
        except Exception as e:                              # the reference prints and exit(1)s (:104-106); raise instead
            raise IOError("load image error %s" % path) from e

    def _depth(self, scene_id, image_id):
        from PIL import Image
        path = self._path(scene_id, "depth", image_id + ".png")
        depth = np.float32(np.asarray(Image.open(path)) / 1000.0)
        depth[depth < 0.1] = 0
        depth[depth > self.depth_scale] = 0
        if not np.max(depth) > 0.0:                          # :120-121
            raise ValueError("depth error %s: no pixel inside (0.1, %g] m" % (path, self.depth_scale))
        return depth

    def _normal(self, scene_id, image_id):
        path = self._path(scene_id, "lg_normal", image_id + ".npy")
        if os.path.exists(path):
            normal = np.float32(np.load(path))
        else:                                               # 16-bit RGB png, (v / 65535 - 0.5) * 2  (:82-88)
            normal = _read_png16_rgb(path[:-4] + ".png")
            normal = (np.float32(normal) / 65535.0 - 0.5) * 2
        return np.where(np.isnan(normal), 0, normal)

    def _camera(self, scene_id, image_id):
        return np.float32(load_cam(self._path(scene_id, "cameras", image_id + "_cam.txt")))

    def view_ids(self, image_id):
        """Reference view first, then id + interval * (v - view_num // 2) for v = 0..view_num-1, skipping 0 (:50-56)."""
        ids = [str(image_id)]
        for view in range(self.view_num):
            i = view - self.view_num // 2
            if i != 0:
                ids.append(str(int(image_id) + self.interval * i))
        return ids

    def __getitem__(self, index):
        scene_id, image_id = self.sample_list[index][:2]
        ids = self.view_ids(image_id)
        sample = {"rgbs": np.stack([self._rgb(scene_id, i) for i in ids], axis=0),
                  "depths": np.stack([self._depth(scene_id, i) for i in (ids if self.source_depths else ids[:1])], axis=0),
                  "normals": np.stack([self._normal(scene_id, ids[0])], axis=0),
                  "cameras": np.stack([self._camera(scene_id, i) for i in ids], axis=0),
                  "filenames": ["%s_%s" % (scene_id, i) for i in ids]}
        if self.planes:
            seg = load_seg(self._path(scene_id, "plane_seg", ids[0] + ".png"))
            para = np.load(self._path(scene_id, "plane_para", ids[0] + ".npy"))
            para, seg = process_by_seg(para, seg, scene_id, ids[0])
            para = plane_para_coordinate_exchange(np.array(para, np.float64))
            num = para.shape[0]
            sample.update({"plane_segs": seg[None], "plane_instance_segs": load_plane_instance_seg(seg, num)[None],
                           "plane_nums": np.array([num], np.int64),
                           "normals_from_plane_para": normal_from_plane_para(para, num, seg)[None]})
        if self.transform:
            sample = self.transform(sample)
        return sample


def _read_png16_rgb(path):
    """16-bit RGB png -> [H,W,3] uint16.  PIL has no 16-bit RGB mode; decode the IDAT stream directly."""
    import struct, zlib
    with open(path, "rb") as f:
        data = f.read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n", path
    pos, idat, hdr = 8, [], None
    while pos < len(data):
        n, kind = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        if kind == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif kind == b"IDAT":
            idat.append(body)
        pos += 12 + n
    w, h, depth, ctype, _, _, interlace = hdr
    if (depth, ctype, interlace) != (16, 2, 0):
        raise ValueError("%s: expected a non-interlaced 16-bit RGB png" % path)
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), np.uint8).reshape(h, 1 + w * 6)
    bpp, out = 6, np.zeros((h, w * 6), np.uint8)
    for y in range(h):                                       # undo the per-row png filters
        ft, line = raw[y, 0], raw[y, 1:].astype(np.int32)
        up = out[y - 1].astype(np.int32) if y else np.zeros(w * 6, np.int32)
        if ft == 0:
            cur = line
        elif ft == 2:
            cur = line + up
        else:
            cur = np.zeros(w * 6, np.int32)
            for x in range(w * 6):
                a = cur[x - bpp] if x >= bpp else 0
                b = up[x]
                c = up[x - bpp] if x >= bpp else 0
                if ft == 1:
                    p = a
                elif ft == 3:
                    p = (a + b) >> 1
                else:
                    pa, pb, pc = abs(b - c), abs(a - c), abs(a + b - 2 * c)
                    p = a if (pa <= pb and pa <= pc) else b if pb <= pc else c
                cur[x] = (line[x] + p) & 255
        out[y] = cur & 255
    return out.reshape(h, w, 3, 2).astype(np.uint16) @ np.array([256, 1], np.uint16)


# ------------------------------------------------------------------ transforms
class Resizer:
    """Images bilinear to the image size, depth / normals (/ segmentations) nearest to the depth size, intrinsics scaled
    by the IMAGE scale factors (dataloader_batch.py:242-336)."""

    def __init__(self, image_width_expected=1280, image_height_expected=960, depth_width_expected=320, depth_height_expected=240):
        self.image_width_expected, self.image_height_expected = image_width_expected, image_height_expected
        self.depth_width_expected, self.depth_height_expected = depth_width_expected, depth_height_expected

    def __call__(self, sample):
        rgbs, cams = sample["rgbs"], sample["cameras"]
        scale_x = float(self.image_width_expected) / rgbs[0].shape[1]
        scale_y = float(self.image_height_expected) / rgbs[0].shape[0]
        ih, iw, dh, dw = self.image_height_expected, self.image_width_expected, self.depth_height_expected, self.depth_width_expected
        out = dict(sample)
        out["rgbs"] = np.stack([resize_linear(v, ih, iw) for v in rgbs])                       # float64, as the reference
        out["depths"] = np.stack([resize_nearest(v, dh, dw) for v in sample["depths"]]).astype(np.float64)
        out["normals"] = np.stack([resize_nearest(v, dh, dw) for v in sample["normals"]]).astype(np.float64)
        out["cameras"] = np.stack([scale_camera(c, scale_x, scale_y) for c in cams])
        for key in ("plane_segs", "normals_from_plane_para"):
            if key in sample:
                out[key] = np.stack([resize_nearest(v, dh, dw) for v in sample[key]])
        if "plane_instance_segs" in sample:                                                  # :322-336
            out["plane_instance_segs"] = np.stack([np.stack([resize_nearest(m, dh, dw) for m in v]) for v in sample["plane_instance_segs"]])
        return out


class ToTensor:
    """numpy -> torch with channels first: rgbs / normals [V,3,H,W] float32, depths [V,1,H,W], cameras [V,2,4,4]
    (dataloader_batch.py:339-421); adds `disparities` from the resized depths."""

    def __call__(self, sample):
        import torch
        chw = lambda a: torch.from_numpy(np.ascontiguousarray(np.float32(a).transpose(0, 3, 1, 2)))
        depths = np.float32(sample["depths"])
        out = {"rgbs": chw(sample["rgbs"]), "depths": torch.from_numpy(depths).unsqueeze(1),
               "disparities": torch.from_numpy(np.float32(disparity_from_depth(depths))).unsqueeze(1),
               "normals": chw(sample["normals"]), "cameras": torch.from_numpy(np.float32(sample["cameras"])),
               "filenames": sample["filenames"]}
        if "plane_segs" in sample:
            out["plane_segs"] = torch.from_numpy(np.ascontiguousarray(sample["plane_segs"]).astype(np.uint8))
            out["plane_instance_segs"] = torch.from_numpy(np.ascontiguousarray(sample["plane_instance_segs"]).astype(np.uint8))
            out["plane_nums"] = torch.from_numpy(np.asarray(sample["plane_nums"], np.int64))
            out["normals_from_plane_para"] = chw(sample["normals_from_plane_para"])
        return out


class Compose:
    def __init__(self, transforms):
        self.transforms = list(transforms)

    def __call__(self, sample):
        for t in self.transforms:
            sample = t(sample)
        return sample


def training_loader(list_filepath, root_dir, image_height, image_width, batch_size, shuffle=True, num_workers=0, seed=0,
                    rank=0, world_size=1, **dataset_kw):
    """The reference's `load_dataset('train', 'scannet')` (train.py:39-56): Resizer to one size for images and depth, then
    ToTensor, batched.  One contiguous list shard per rank instead of a DataParallel scatter."""
    import torch
    transform = Compose([Resizer(image_width, image_height, image_width, image_height), ToTensor()])
    ds = ScannetDataset(list_filepath, root_dir, transform=transform, rank=rank, world_size=world_size, **dataset_kw)
    g = torch.Generator(); g.manual_seed(seed)
    return torch.utils.data.DataLoader(ds, batch_size=batch_size, shuffle=shuffle, num_workers=num_workers, drop_last=True, generator=g)


# ------------------------------------------------------------------ a ScanNet-shaped scene without the dataset
def write_synthetic_scene(root_dir, scene_id="scene0000_00", num_frames=5, interval=10, height=48, width=64, seed=0,
                          rgb_ext=".jpg", planes=False, list_name="list.txt"):
    """A slanted textured floor + back wall seen from a camera translating sideways: rgb / depth / lg_normal / cameras
    (and plane_seg / plane_para) under root_dir/scene_id, ids 0, interval, 2*interval, ...; the list file names every id
    that has both neighbours.  Returns the list path."""
    from PIL import Image
    rng = np.random.default_rng(seed)
    scene = os.path.join(root_dir, scene_id)
    for sub in ("rgb", "depth", "lg_normal", "cameras") + (("plane_seg", "plane_para") if planes else ()):
        os.makedirs(os.path.join(scene, sub), exist_ok=True)
    K = np.array([[1.1 * width, 0, width / 2.0], [0, 1.4 * height, height / 2.0], [0, 0, 1]])
    tex = rng.integers(0, 255, (height // 4 + 2, width // 4 + 2 * num_frames + 2, 3)).astype(np.float64)
    ys, xs = np.mgrid[0:height, 0:width].astype(np.float64)
    horizon = height // 2
    depth = np.where(ys < horizon, 3.0, 3.0 - 2.2 * (ys - horizon) / (height - horizon))      # wall, then a floor rising to 0.8 m
    normal = np.zeros((height, width, 3), np.float32)
    normal[ys < horizon] = (0, 0, -1); normal[ys >= horizon] = (0, -0.8, -0.6)
    for f in range(num_frames):
        name = str(f * interval)
        big = resize_linear(tex, tex.shape[0] * 4, tex.shape[1] * 4)
        rgb = np.clip(big[4:4 + height, 4 + 2 * f:4 + 2 * f + width], 0, 255).astype(np.uint8)
        Image.fromarray(rgb, "RGB").save(os.path.join(scene, "rgb", name + rgb_ext), **({"quality": 95} if rgb_ext == ".jpg" else {}))
        Image.fromarray((depth * 1000).astype(np.uint16)).save(os.path.join(scene, "depth", name + ".png"))
        nrm = normal.copy(); nrm[0, 0] = np.nan
        np.save(os.path.join(scene, "lg_normal", name + ".npy"), nrm)
        ext = np.eye(4); ext[0, 3] = -0.05 * f
        write_cam(os.path.join(scene, "cameras", name + "_cam.txt"), ext, K)
        if planes:
            seg = np.where(ys < horizon, 3, 7).astype(np.uint8); seg[:, :2] = 9            # labels 3 and 7 planar, 9 = largest = non-planar
            Image.fromarray(seg).save(os.path.join(scene, "plane_seg", name + ".png"))
            para = np.zeros((10, 3)); para[3] = (0, -1 * 3.0, 0); para[7] = (0, -0.6, 0.8)
            np.save(os.path.join(scene, "plane_para", name + ".npy"), para)
    list_path = os.path.join(root_dir, list_name)
    with open(list_path, "w") as f:
        for i in range(1, num_frames - 1):
            f.write("%s %d\n" % (scene_id, i * interval))
    return list_path
