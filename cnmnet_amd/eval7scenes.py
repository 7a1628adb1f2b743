"""7-Scenes evaluation harness around the engine's frame pipeline.

Host-side restatement of the reference's evaluation protocol:
  * sequence list, file naming, intrinsics, pose -> extrinsics, camera scaling, image normalisation
    (reference eval.py:26-159, class LoadSevenScenes),
  * which frames form a sample for 2 / 3 / 5 / 7 views (eval.py:239-245, :408-415, :581-592, :817-830),
  * how a prediction is written out (eval.py:497-501: depth = 1 / (idepth + 1e-4), > 100 m -> 0),
  * the per-frame depth errors and their aggregation (eval.py:996-1090 cal_metrics, utils/metric.py:149-445).
Image files are read with PIL (the reference uses cv2, absent here); resizing restates cv2.INTER_LINEAR's float path
(half-pixel centres, edge clamp, no antialiasing) -- parity unpinned for uint8 inputs, where cv2 uses 11-bit fixed-point
weights (differences <= 1 grey level).  The depth metrics are pinned to the imported reference by
tests/golden/metrics.npz (tests/test_eval7scenes_cpu.py).
"""
import os

import numpy as np

TEST_SEQUENCES = (("chess", "seq-03"), ("chess", "seq-05"), ("fire", "seq-03"), ("fire", "seq-04"), ("heads", "seq-01"),
                  ("office", "seq-02"), ("office", "seq-06"), ("office", "seq-07"), ("office", "seq-09"),
                  ("pumpkin", "seq-01"), ("pumpkin", "seq-07"), ("redkitchen", "seq-03"), ("redkitchen", "seq-04"),
                  ("redkitchen", "seq-06"), ("redkitchen", "seq-12"), ("redkitchen", "seq-14"), ("stairs", "seq-01"),
                  ("stairs", "seq-04"))                                   # eval.py:29-46
INTRINSICS = np.array([[585.0, 0, 320], [0, 585.0, 240], [0, 0, 1]])    # 640x480 Kinect, eval.py:48-50
IMAGENET_MEAN, IMAGENET_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
MIN_DEPTH, MAX_DEPTH = 0.3, 8.0                                          # eval.py:1009-1010, utils/metric.py:149

# source-view offsets (in frames), and the reference-frame subsampling per number of views
_OFFSETS = {2: (10,), 3: (10, -10), 5: (10, -10, 5, -5), 7: (10, -10, 5, -5, 20, -20)}
_STRIDE = {2: 10, 3: 3, 5: 3, 7: 9}                                      # eval.py:240, :409, :582, :818 (`if index % k != 0: continue`)


def sample_indices(num_frames, views):
    """[(ref, [sources...])] exactly as the reference enumerates AND subsamples them: eval.py:239-240 (2 views:
    range(0, n-10), every 10th frame), :408-409 (3 views: range(10, n-10), every 3rd), :581-582 (5 views:
    range(10, n-20), every 3rd), :817-818 (7 views: range(10, n-20), every 9th).
    With 7 views the reference's first sample is frame 18, whose -20 source is index -2: Python wraps it to the
    second-to-last frame of the sequence, and so does this list (kept for parity of the evaluated sample set; such
    a pair has no overlap and simply contributes an uninformative source)."""
    off, k = _OFFSETS[views], _STRIDE[views]
    lo, hi = (0, num_frames - 10) if views == 2 else (10, num_frames - 10) if views == 3 else (10, num_frames - 20)
    return [(i, [i + o for o in off]) for i in range(lo, hi) if i % k == 0]


def sequence_files(seq_dir):
    """Sorted samples of one sequence directory (eval.py:52-70): every '*color*' file with its depth / pose siblings."""
    out = []
    for name in sorted(os.listdir(seq_dir)):
        if "color" in name:
            out.append({"rgb": os.path.join(seq_dir, name), "depth": os.path.join(seq_dir, name.replace("color", "depth")),
                        "pose": os.path.join(seq_dir, name.replace("color.png", "pose.txt")),
                        "pred_depth_name": name.replace("color", "pred_depth")})
    return out


def make_cam(pose_cam_to_world, intrinsics=INTRINSICS, scale_x=1.0, scale_y=1.0):
    """cam [2,4,4]: [0] = world->camera extrinsic = inverse pose (eval.py:127-130), [1][:3,:3] = intrinsics with focal
    lengths and principal point scaled to the network resolution (eval.py:132-159)."""
    cam = np.zeros((2, 4, 4), np.float64)
    cam[0] = np.linalg.inv(np.asarray(pose_cam_to_world, np.float64))
    K = np.array(intrinsics, np.float64)
    K[0, 0] *= scale_x; K[0, 2] *= scale_x
    K[1, 1] *= scale_y; K[1, 2] *= scale_y
    cam[1, :3, :3] = K
    return cam.astype(np.float32)


def normalize_image(rgb):
    """uint8 / float RGB [H,W,3] -> zero-mean unit-variance float32 (eval.py:109-119)."""
    x = np.asarray(rgb, np.float64) / 255.0
    return ((x - IMAGENET_MEAN) / IMAGENET_STD).astype(np.float32)


def resize_linear(img, out_h, out_w):
    """cv2.resize(..., INTER_LINEAR) in floating point: src = (dst + 0.5) * scale - 0.5, clamped to the image."""
    img = np.asarray(img)
    h, w = img.shape[:2]
    def axis(n_out, n_in):
        s = (np.arange(n_out, dtype=np.float64) + 0.5) * (n_in / n_out) - 0.5
        i0 = np.floor(s).astype(np.int64)
        f = s - i0
        lo, hi = np.clip(i0, 0, n_in - 1), np.clip(i0 + 1, 0, n_in - 1)
        return lo, hi, f
    y0, y1, fy = axis(out_h, h)
    x0, x1, fx = axis(out_w, w)
    a = img.astype(np.float64)
    fy = fy.reshape(-1, *([1] * (a.ndim - 1))); fxs = fx.reshape(1, -1, *([1] * (a.ndim - 2)))
    rows = a[y0] * (1 - fy) + a[y1] * fy
    return rows[:, x0] * (1 - fxs) + rows[:, x1] * fxs


def read_pose(path):
    return np.loadtxt(path, dtype=np.float32).reshape(4, 4)              # eval.py:121-125 (tab/space separated 4x4)


def load_sample(files, height, width):
    """(rgb [3,H,W] float32 normalised, depth [h0,w0] float32 metres at the ORIGINAL resolution, cam [2,4,4]) -- eval.py:72-94."""
    from PIL import Image
    rgb = np.asarray(Image.open(files["rgb"]).convert("RGB"))
    depth = np.asarray(Image.open(files["depth"])).astype(np.float64) / 1000.0
    h0, w0 = rgb.shape[:2]
    cam = make_cam(read_pose(files["pose"]), INTRINSICS, width / float(w0), height / float(h0))
    # cv2.resize on a uint8 image returns uint8 (rounded, saturated) and the reference normalises THAT (eval.py:84-90);
    # rounding here keeps the network input on the same 8-bit grid (cv2's fixed-point weights can still differ by 1 level)
    rgb = np.clip(np.rint(resize_linear(rgb, height, width)), 0, 255).astype(np.uint8)
    return np.ascontiguousarray(normalize_image(rgb).transpose(2, 0, 1)), depth.astype(np.float32), cam


def depth_from_idepth(idepth):
    """What the reference saves as pred_depth.npy (eval.py:497-499)."""
    d = np.reciprocal(np.asarray(idepth, np.float64) + 1e-4)
    d[d > 100] = 0
    return d.astype(np.float32)


# ------------------------------------------------------------------ depth error measures (utils/metric.py:149-445)
def valid_depth_mask(d1, d2=None, min_thred=MIN_DEPTH, max_thred=MAX_DEPTH):
    """utils/metric.py:149-162.  One map: inside (min, max) and finite; two maps: both inside (min, max)."""
    d1 = np.asarray(d1)
    if d2 is None:
        return (d1 < max_thred) & (d1 > min_thred) & np.isfinite(d1)
    d2 = np.asarray(d2)
    return (d1 < max_thred) & (d2 < max_thred) & (d1 > min_thred) & (d2 > min_thred)


def _pair(a, b):
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    if not (np.all(np.isfinite(a) & np.isfinite(b)) and np.all((a > 0) & (b > 0))):
        raise AssertionError("depth error measures take finite, positive depths (mask first)")     # the reference asserts
    return a, b


def distances(depth_pred, depth_gt, thresholds=(1.25, 1.25 ** 2, 1.25 ** 3)):
    """All measures of utils/metric.py:165-362 on already masked depths; NaN for empty input.
    Argument roles follow the reference: relative errors divide by depth_gt."""
    p, g = _pair(depth_pred, depth_gt)
    n = float(p.size)
    if n == 0:
        return {k: np.nan for k in ("l1", "l1_inverse", "rmse_log", "rmse", "scale_invariant", "abs_relative", "avg_log10",
                                    "sq_relative") + tuple("ratio_threshold_%s" % t for t in thresholds)}
    diff, ldiff = p - g, np.log(p) - np.log(g)
    out = {"l1": np.abs(diff).sum() / n,
           "l1_inverse": np.abs(1.0 / p - 1.0 / g).sum() / n,
           "rmse_log": np.sqrt(np.square(ldiff).sum() / n),
           "rmse": np.sqrt(np.square(diff).sum() / n),
           "scale_invariant": np.sqrt(np.square(ldiff).sum() / n - np.square(ldiff.sum()) / (n * n)),
           "abs_relative": (np.abs(diff) / g).sum() / n,
           "avg_log10": np.abs(np.log10(p) - np.log10(g)).sum() / n,
           "sq_relative": (np.square(diff) / g).sum() / n}
    for t in thresholds:
        out["ratio_threshold_%s" % t] = float((np.abs(ldiff) < np.log(t)).sum()) / n
    return out


def compute_errors(depth_pred, depth_gt):
    """utils/metric.py:365-404: mask where BOTH maps are in (0.3, 8) m, then every distance; 'num_valid' included."""
    m = valid_depth_mask(depth_pred, depth_gt)
    out = distances(np.asarray(depth_pred)[m], np.asarray(depth_gt)[m])
    out["num_valid"] = int(m.sum())
    return out


def depth_scale_factor(depth1, depth2, depth_scaling="abs"):
    """Least-squares scale of depth1 towards depth2 (utils/metric.py:407-445)."""
    a, b = _pair(depth1, depth2)
    if depth_scaling == "log":
        return float(np.exp(np.mean(np.log(b) - np.log(a))))
    if depth_scaling == "abs":
        aa, ab = a * a, a * b
    elif depth_scaling == "inv":
        aa, ab = 1.0 / (a * a), 1.0 / (a * b)
    else:
        raise ValueError("unknown depth scaling method %r" % (depth_scaling,))
    m = valid_depth_mask(ab)                                             # the reference masks on the PRODUCT map (:419, :434)
    saa, sab = aa[m].sum(), ab[m].sum()
    if not saa > 0:
        return 1.0
    return float(sab / saa) if depth_scaling == "abs" else float(saa / sab)


def frame_errors(gt_depth, pred_depth):
    """Per-frame numbers of cal_metrics (eval.py:1023-1045): prediction resized to the ground-truth resolution,
    clamped to [0.3, 8] m, evaluated where the GROUND TRUTH is valid.  Key names = the labels the reference prints."""
    gt = np.asarray(gt_depth, np.float64)
    pred = np.asarray(pred_depth, np.float64)
    if pred.shape != gt.shape:
        pred = resize_linear(pred, *gt.shape)
    pred = np.clip(pred, MIN_DEPTH, MAX_DEPTH)
    m = valid_depth_mask(gt)
    d = distances(pred[m], gt[m])
    return {"mean_l1_error": d["l1"], "a<1.25": d["ratio_threshold_1.25"], "a<1.25^2": d["ratio_threshold_%s" % (1.25 ** 2)],
            "a<1.25^3": d["ratio_threshold_%s" % (1.25 ** 3)], "abs.rel": d["abs_relative"], "sq.rel": d["sq_relative"],
            "rmse": d["rmse"], "rmse_log": d["rmse_log"], "scale.inv": d["scale_invariant"]}


def aggregate(per_frame):
    """Mean over frames of every per-frame error (eval.py:1063-1072)."""
    keys = per_frame[0].keys()
    return {k: float(np.mean([f[k] for f in per_frame])) for k in keys}


# ------------------------------------------------------------------ running the engine over a sequence
def evaluate_sequence(pipeline, seq_dir, height, width, views=3, batch=8, device="cuda:0", limit=None):
    """Frames of one sequence through cnmnet_amd.pipeline.FramePipeline (views = 3, 5 or 7: reference + 2/4/6 sources),
    `batch` frames per engine call.  Returns (per-frame error dicts, aggregate)."""
    import torch
    files = sequence_files(seq_dir)
    samples = sample_indices(len(files), views)
    if limit is not None:
        samples = samples[:limit]
    cache, errors = {}, []
    def get(i):
        if i not in cache:
            cache[i] = load_sample(files[i], height, width)
        return cache[i]
    for s in range(0, len(samples), batch):
        chunk = samples[s:s + batch]
        imgs = np.stack([np.stack([get(i)[0] for i in [r] + src]) for r, src in chunk])
        cams = np.stack([np.stack([get(i)[2] for i in [r] + src]) for r, src in chunk])
        out = pipeline(torch.from_numpy(imgs).to(device), torch.from_numpy(cams).to(device))
        idepth = out["disp"].float().cpu().numpy().reshape(len(chunk), height, width)
        for (r, _), idp in zip(chunk, idepth):
            errors.append(frame_errors(get(r)[1], depth_from_idepth(idp)))
        for k in [k for k in cache if k < chunk[-1][0] - 25]:
            del cache[k]
    return errors, aggregate(errors)


def write_synthetic_sequence(seq_dir, num_frames=40, height=480, width=640, seed=0):
    """A small 7-Scenes-shaped sequence (frame-%06d.color.png / .depth.png / .pose.txt) of a textured fronto-parallel
    wall seen from a camera translating sideways -- for exercising the harness without the dataset."""
    from PIL import Image
    os.makedirs(seq_dir, exist_ok=True)
    rng = np.random.default_rng(seed)
    tex = rng.integers(0, 255, (height // 8 + 2, width // 8 + 40, 3)).astype(np.float64)
    wall_z = 2.0
    for f in range(num_frames):
        tx = 0.01 * f
        shift = INTRINSICS[0, 0] * tx / wall_z                          # pixels the wall moves against the camera
        big = resize_linear(tex, (height // 8 + 2) * 8, (width // 8 + 40) * 8)
        x0 = int(round(shift)) + 8
        rgb = np.clip(big[8:8 + height, x0:x0 + width], 0, 255).astype(np.uint8)
        Image.fromarray(rgb, "RGB").save(os.path.join(seq_dir, "frame-%06d.color.png" % f))
        depth = np.full((height, width), int(wall_z * 1000), np.uint16)
        Image.fromarray(depth).save(os.path.join(seq_dir, "frame-%06d.depth.png" % f))
        pose = np.eye(4); pose[0, 3] = tx
        np.savetxt(os.path.join(seq_dir, "frame-%06d.pose.txt" % f), pose, fmt="%.7e", delimiter="\t ")


def load_checkpoint(path_or_dict, depth_net, refine_net=None):
    """Reference checkpoints (eval.py:181-197, :340-362): keys 'depth_network_state_dict' / 'depth_refine_network_state_dict'
    (fallback 'state_dict' for the depth net), parameter names optionally prefixed 'module.' by nn.DataParallel."""
    import torch
    ck = torch.load(path_or_dict, map_location="cpu") if isinstance(path_or_dict, (str, bytes, os.PathLike)) else path_or_dict
    strip = lambda sd: {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}
    depth_net.load_state_dict(strip(ck["depth_network_state_dict"] if "depth_network_state_dict" in ck else ck["state_dict"]))
    if refine_net is not None:
        refine_net.load_state_dict(strip(ck["depth_refine_network_state_dict"]))
    return depth_net, refine_net
