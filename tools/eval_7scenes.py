#!/usr/bin/env python3
"""Score a CNMNet checkpoint on 7-Scenes with the engine (protocol of the reference's eval.py eval_refine* + cal_metrics).

    python tools/eval_7scenes.py --root /data/7scenes --checkpoint model.pth.tar [--views 3|5|7] [--height 192 --width 256]
    python tools/eval_7scenes.py --synthetic /tmp/fake7scenes            # writes a tiny fake dataset first (no checkpoint)
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cnmnet_amd import eval7scenes as e7
from cnmnet_amd.depthnet import depthNet, DepthRefineNet
from cnmnet_amd.pipeline import FramePipeline


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--root"); ap.add_argument("--synthetic"); ap.add_argument("--checkpoint")
    ap.add_argument("--views", type=int, default=3, choices=[3, 5, 7])
    ap.add_argument("--height", type=int, default=192); ap.add_argument("--width", type=int, default=256)
    ap.add_argument("--idepth-scale", type=float, default=3.0); ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--limit", type=int, default=None, help="frames per sequence")
    a = ap.parse_args()
    root, seqs = a.root, e7.TEST_SEQUENCES
    if a.synthetic:
        root, seqs = a.synthetic, (("chess", "seq-03"),)
        e7.write_synthetic_sequence(os.path.join(root, "chess", "seq-03"), num_frames=48, seed=1)
    dev = torch.device("cuda:0")
    dn, rn = depthNet(a.idepth_scale).to(dev).eval(), DepthRefineNet(32, a.idepth_scale).to(dev).eval()
    if a.checkpoint:
        e7.load_checkpoint(a.checkpoint, dn, rn)
    pipe = FramePipeline(dn, rn, k_size=9, normals=False)
    per_frame, t0 = [], time.perf_counter()
    for scene, seq in seqs:
        d = os.path.join(root, scene, seq)
        if not os.path.isdir(d):
            print("skip (missing):", d); continue
        errs, agg = e7.evaluate_sequence(pipe, d, a.height, a.width, views=a.views, batch=a.batch, device=dev, limit=a.limit)
        per_frame += errs
        print(scene, seq, len(errs), "frames", json.dumps({k: round(v, 4) for k, v in agg.items()}))
    if per_frame:
        print("ALL", len(per_frame), "frames in %.1f s" % (time.perf_counter() - t0), json.dumps(e7.aggregate(per_frame)))


if __name__ == "__main__":
    main()
