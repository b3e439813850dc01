"""Build libimcui_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.normpath(os.path.join(PKG_DIR, "..", "csrc"))
INCLUDE = os.path.normpath(os.path.join(PKG_DIR, "..", "..", "include"))
LIB_DIR = os.path.join(PKG_DIR, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libimcui_hip.so")
SOURCES = ["api.hip", "preprocess.hip", "jpeg.hip", "png.hip", "geometry.hip", "gemm.hip", "gemm_wreg.hip", "ffn.hip", "simred.hip", "conv.hip", "attention.hip", "attention_mx.hip", "superpoint.hip", "lightglue.hip", "superglue.hip", "nn.hip", "dual_softmax.hip", "loftr.hip", "eloftr.hip", "dust3r.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", f"-I{INCLUDE}", f"-I{CSRC}"]
# Per-source code-generation switches.  preprocess.hip restates host float32 arithmetic that rounds after every
# multiply and every add: hipcc's default FMA contraction would change the last bit.
# gemm_wreg.hip: no SLP vectorisation -- hipcc packed the four rotary rotations of a lane into v_pk_fma_f32 / v_pk_mul_f32 with
# op_sel swizzles and in-place destinations, and on gfx950 that sequence intermittently returned wrong even elements in lanes
# 48-63 (found with tools/r03_diag2.py in round 3; scalar fused multiply-adds are bit-stable).
# gemm.hip (round 4): the same switch for the round-2 GEMM, whose rotary epilogue (EPI_QKV, taken when gemm_wreg_ok says no) compiled to the
# same class of in-place v_pk_fma_f32 with op_sel (16 per kernel, seen in the ISA); measured performance-neutral on one box (headline
# 945.6 -> 950.2 pairs/s, LoFTR 100.98 -> 101.04, DUSt3R 89.98 -> 89.83), all kernel tests unchanged.
EXTRA_FLAGS: dict = {"preprocess.hip": ["-ffp-contract=off"], "gemm_wreg.hip": ["-fno-slp-vectorize"], "gemm.hip": ["-fno-slp-vectorize"]}


def _newest_source_mtime() -> float:
    files = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(INCLUDE, "imcui_hip.h"), os.path.abspath(__file__)]  # (the flags live in this file)
    return max(os.path.getmtime(f) for f in files)


def needs_build() -> bool:
    return not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < _newest_source_mtime()


# MFMA kernels whose register budget is part of the design: a build that spills them to scratch is rejected (two
# experimental builds of the split GEMM that spilled -- 128-VGPR and 168-VGPR-with-20-B-scratch variants -- were
# slower AND failed the parity / determinism tests on the GPU; hipcc is not to be trusted with spills around them).
NO_SPILL_KERNELS = ("simred_kernel", "gemm_split_kernel", "gemm_wreg_kernel", "gemm_kernel", "attn_split_kernel", "attn_mx_kernel", "attn_kernel", "conv3x3_split_kernel", "conv3x3_tall_kernel", "conv3x3_kernel", "lg_ffn_kernel")


def _check_no_spills(src: str, remarks: str) -> None:
    name = None
    for line in remarks.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
        m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
        if m and name and int(m.group(1)) > 0 and any(k in name for k in NO_SPILL_KERNELS):
            raise RuntimeError(f"{src}: kernel {name} spills {m.group(1)} bytes/lane to scratch -- reduce its register pressure")


def build(force: bool = False, verbose: bool = True) -> str:
    """Compile every HIP source and link the shared library. Returns its path.

    Safe to call from several processes at once (the ranks of one `torch.distributed.run` launch on a clean checkout, pytest-xdist
    workers): the build runs under an exclusive `flock` on lib/.build.lock, the freshness test is repeated once the lock is held, and
    the library is linked under a temporary name and renamed into place, so a concurrent `dlopen` never sees a half-written file."""
    if not force and not needs_build():
        return LIB_PATH
    import fcntl

    os.makedirs(LIB_DIR, exist_ok=True)
    with open(os.path.join(LIB_DIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if force or needs_build():  # (another process may have finished the build while this one waited)
                _build_locked(verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB_PATH


def _build_locked(verbose: bool) -> None:
    obj_dir = os.path.join(LIB_DIR, "obj")
    os.makedirs(obj_dir, exist_ok=True)

    def compile_one(src: str) -> str:
        obj = os.path.join(obj_dir, src.replace(".hip", ".o"))
        cmd = [HIPCC, *FLAGS, *EXTRA_FLAGS.get(src, []), "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr}")
        _check_no_spills(src, r.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    tmp = LIB_PATH + f".tmp{os.getpid()}"
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-lz", "-o", tmp]  # (-lz: the PNG path inflates on host threads with the system zlib)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr}")
    os.replace(tmp, LIB_PATH)
    if verbose:
        print(f"[imcui_hip] built {LIB_PATH}", file=sys.stderr)


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
