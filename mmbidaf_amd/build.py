"""In-tree build of libmmbidaf_hip.so with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
SOURCES = ["api.hip", "gemm.hip", "gemm_bf16.hip", "planes.hip", "lstm.hip", "lstm_big.hip", "bidaf.hip", "bidaf_big.hip", "decoder.hip", "highway.hip"]
HEADERS = ["common.h", os.path.join("..", "..", "include", "mmbidaf.h")]
LIB = os.path.join(_HERE, "libmmbidaf_hip.so")


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -shared ... -> mmbidaf_amd/libmmbidaf_hip.so"""
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-shared", "-fPIC",
           "-Wno-unused-result"] + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True, cwd=CSRC)
    return LIB


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
