"""Build the gfx950 engine library in-tree:  python -m cnmnet_amd.build

hipcc cross-compiles without a GPU; the resulting cnmnet_amd/lib/libcnm_engine.so is
git-ignored but travels to the GPU box with the working tree.
"""
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
LIB = os.path.join(LIBDIR, "libcnm_engine.so")
SOURCES = ["planesweep.hip", "conv_mfma.hip", "conv_winograd.hip", "conv_winograd4.hip", "conv_winograd4s.hip", "conv_winograd4q.hip", "conv_winograd_rows.hip", "conv_rows_staged.hip", "pointwise.hip", "geometry.hip", "nets.hip", "train_ops.hip", "half_ops.hip"]
HOST_SOURCES = ["host_twins.cpp"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
HOSTCXX = os.environ.get("CXX", "g++")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
# per-file extras: the plane sweep's blend must stay on full-rate scalar v_fma_f32 (the SLP vectoriser would
# pack it into half-rate v_pk_fma_f32 plus the moves that feed them)
# the staged 36-point kernel lives at the register limit (256 VGPRs, 106 SGPRs + spill lanes): the greedy allocator's class-priority
# order leaves the dominant instance without VGPR spills, and without machine LICM fewer scalars are kept live across the phase loop
# (67 instead of 71 v_readlane reloads per phase, tools/hotloop_proxy.sh; 5.52 vs 5.57 ms over a step's launches, tools/r4_wino_variants.sh).
# CNM_NO_FILE_FLAGS=1 builds without them (A/B).
# the fp16 heads (half_ops.hip) multiply half inputs by fp32 weights: left alone, the SLP vectoriser packs the products into v_pk_fma_f32 (which
# issues as two FMAs) behind 240 v_cvt_f32_f16 per channel group; without it the compiler selects v_fma_mix_f32 (conversion folded in): 656
# instead of 896 VALU instructions per group and lane.
FILE_FLAGS = {"planesweep.hip": ["-fno-slp-vectorize"],
              "half_ops.hip": ["-fno-slp-vectorize"],
              "conv_winograd4s.hip": ["-mllvm", "-greedy-regclass-priority-trumps-globalness=1", "-mllvm", "-disable-machine-licm"],
              "conv_rows_staged.hip": ["-mllvm", "-greedy-regclass-priority-trumps-globalness=1", "-mllvm", "-disable-machine-licm"]}   # 37 instead of 60 spilled SGPRs
if os.environ.get("CNM_NO_FILE_FLAGS") == "1":
    FILE_FLAGS = {"planesweep.hip": ["-fno-slp-vectorize"], "half_ops.hip": ["-fno-slp-vectorize"]}


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    os.makedirs(LIBDIR, exist_ok=True)
    headers = [os.path.join(CSRC, "cnm_common.h"), os.path.join(CSRC, "wino4_args.h"), os.path.join(CSRC, "rows_args.h"), os.path.join(CSRC, "sync_ws.h"), os.path.join(CSRC, "host_ops.h"), os.path.join(PKG, "..", "include", "cnm_engine.h")]
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(LIBDIR, src.replace(".hip", ".o"))
        if force or _stale(o, [s, os.path.abspath(__file__)] + headers):
            cmd = [HIPCC] + FLAGS + FILE_FLAGS.get(src, []) + os.environ.get("CNM_EXTRA_HIPCC_FLAGS", "").split() + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        objs.append(o)
    # host twins (plain C++, no HIP): g++ for the AVX2 / FMA function clones, resolved at load time by the CPU that runs them
    for src in HOST_SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(LIBDIR, src.replace(".cpp", ".o"))
        if force or _stale(o, [s, os.path.abspath(__file__), os.path.join(CSRC, "host_ops.h"), os.path.join(PKG, "..", "include", "cnm_engine.h")]):
            cmd = [HOSTCXX, "-O3", "-std=c++17", "-fPIC", "-Wall", "-pthread", "-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        objs.append(o)
    if force or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread"] + objs + ["-o", LIB]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    # A/B twin for bench.py's counter pass: the same library with the staged 36-point kernel's units ordered channel
    # block FASTEST (-DWINO4S_CBLK_SLOW=0).  Never loaded by the product (cnmnet_amd/_lib.py loads it only when CNM_ENGINE_LIB names it).
    s = os.path.join(CSRC, "conv_winograd4s.hip")
    o = os.path.join(LIBDIR, "conv_winograd4s_cblk0.o")
    alt = os.path.join(LIBDIR, "libcnm_engine_cblk0.so")
    if force or _stale(o, [s, os.path.abspath(__file__)] + headers):
        cmd = [HIPCC] + FLAGS + FILE_FLAGS.get("conv_winograd4s.hip", []) + ["-DWINO4S_CBLK_SLOW=0", "-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    alt_objs = [o if x.endswith("conv_winograd4s.o") else x for x in objs]
    if force or _stale(alt, alt_objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread"] + alt_objs + ["-o", alt]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
