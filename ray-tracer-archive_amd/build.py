"""Builds librt_hip.so (HIP kernels + C ABI + host mirror) for gfx950 with hipcc, in-tree."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "librt_hip.so")
SOURCES = ["csrc/kernels.hip", "csrc/rt_api.cpp", "csrc/rt_multi.cpp", "csrc/scene_compile.cpp", "csrc/wide_bvh.cpp", "host/host_capi.cpp"]
HEADERS = ["csrc/kernels.h", "csrc/device_types.h", "csrc/rt_internal.hpp", "csrc/scene_compile.hpp", "csrc/wide_bvh.hpp", "host/rt_host.hpp", "host/jpeg_writer.hpp", "../include/rt_hip.h", "../include/rt_host.h"]
# -ffp-contract=off: a float expression means the same IEEE operations wherever it is inlined, so a
# sample's radiance does not depend on which kernel / call site generated its camera ray (and the
# device evaluates the reference's expressions in the order written). Hot loops spell out fmaf/fma.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function", "-ffp-contract=off",
         # every atomic here is already one-per-wave / one-per-workgroup; LLVM's wave-aggregation pass would only add a
         # readfirstlane that forces an immediate wait on the returning atomic (k_extend grabs its chunk one ahead)
         "-mllvm", "-amdgpu-atomic-optimizer-strategy=None"]


def source_hash():
    """sha256 over the sources that decide what the kernels execute per segment — the kernels, and the scene compiler and uploader that
    lay the scene out for them: what a PMC profile under profiles/ is valid for (bench.py refuses a profile taken from other sources)."""
    import hashlib
    h = hashlib.sha256()
    for f in ["csrc/kernels.hip", "csrc/kernels.h", "csrc/device_types.h", "csrc/scene_compile.cpp", "csrc/rt_api.cpp", "csrc/wide_bvh.cpp"]:
        h.update(open(os.path.join(HERE, f), "rb").read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()[:16]


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(os.path.join(HERE, f)) > t for f in SOURCES + HEADERS)


def build_variant(name, defines, verbose=False):
    """Tuning builds: lib/variants/librt_hip_<name>.so with extra -D flags (scripts/ only)."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    vdir = os.path.join(LIB_DIR, "variants")
    os.makedirs(vdir, exist_ok=True)
    out = os.path.join(vdir, f"librt_hip_{name}.so")
    cmd = [hipcc] + FLAGS + [f"-D{d}" for d in defines] + ["-o", out] + SOURCES + ["-lz", "-ldl"]
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, cwd=HERE, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + r.stdout + r.stderr)
    return out


def build(force=False, verbose=False):
    """Compile every HIP source of the package for gfx950. Raises on failure."""
    if not force and not needs_build():
        return LIB_PATH
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build the HIP library")
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [hipcc] + FLAGS + ["-o", LIB_PATH] + SOURCES + ["-lz", "-ldl"]
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, cwd=HERE, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + r.stdout + r.stderr)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
