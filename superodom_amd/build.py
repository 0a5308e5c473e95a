"""Build libsoicp.so (HIP kernels + C++ host driver + C ABI) for gfx950, in-tree.

    python -m superodom_amd.build [--force]

hipcc cross-compiles without a GPU; the resulting superodom_amd/lib/libsoicp.so is git-ignored but
travels with the gpurun snapshot."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libsoicp.so")
SOURCES = ["kernels.hip", "map_kernels.hip", "icp_context.cpp", "local_map.cpp", "device_map.cpp"]
HEADERS = ["kernels.h", "map_kernels.h", "device_map.h", "lm_solver.h", "local_map.h", "so_math.h", "deskew_math.h", "plane_fit.h", os.path.join("..", "..", "include", "so_icp.h")]
ARCH = "gfx950"
# -ffp-contract=off: the fp64 plane fit / evaluation follow the reference's unfused arithmetic
COMMON = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function", "-Wno-unused-result"]
COMMON += os.environ.get("SOICP_EXTRA_CXXFLAGS", "").split()  # experiments, e.g. -DSO_SOLVE_BLOCKS=512


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=False):
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(LIBDIR, exist_ok=True)
    objs = []
    for s in SOURCES:
        o = os.path.join(LIBDIR, os.path.splitext(s)[0] + ".o")
        lang = ["-x", "hip"] if s.endswith(".hip") else []  # host-only .cpp files are plain C++
        cmd = [hipcc, f"--offload-arch={ARCH}"] + lang + ["-c", os.path.join(CSRC, s), "-o", o] + COMMON
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        objs.append(o)
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-o", LIB] + objs + ["-ldl"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
