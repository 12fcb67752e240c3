"""Builds libds2i_hip.so (HIP kernels + C-ABI + host index builder) in-tree with hipcc for gfx950.

The shared library is the product's only compute path; there is no CPU fallback.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libds2i_hip.so")
ARCH = "gfx950"

# ranked_stream.hip holds the pipelined ranked_and kernels of the benchmark configuration;
# kernels.hip is compiled six times: once per list-count class (-DDS2I_TU_TMAX=n: the query kernels of that class; 0 = the
# long class) and once for everything else; encode_kernels.hip holds the index encoder; all units are built in parallel
DEVICE_UNITS = [("kernels.hip", "kernels_t%d.hip" % t, ["-DDS2I_TU_TMAX=%d" % t]) for t in (2, 4, 8, 16, 0)] + [("kernels.hip", "kernels.hip", []), ("ranked_stream.hip", "ranked_stream.hip", []), ("ranked_stream.hip", "ranked_stream_bigk.hip", ["-DDS2I_RS_BIGK_TU"]), ("ranked_stream_mixed.hip", "ranked_stream_mixed.hip", []), ("freq_stream.hip", "freq_stream.hip", []), ("union_stream.hip", "union_stream.hip", []), ("union_stream.hip", "union_stream_bigk.hip", ["-DDS2I_US_BIGK_TU"]), ("encode_kernels.hip", "encode_kernels.hip", [])]
HOST_SRCS = ["capi.cpp", "capi_batch.cpp", "capi_build.cpp", "capi_encode.cpp"]
COMMON = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function",
          "-Wno-unused-variable", "-Wno-unused-but-set-variable"]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".inc"))]
    inc = os.path.join(HERE, "..", "include")
    hs += [os.path.join(inc, f) for f in os.listdir(inc) if f.endswith(".h")]
    return hs


def build(verbose=False, force=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = os.environ.get("DS2I_EXTRA_CFLAGS", "").split()  # e.g. -DDS2I_PHASE_TIMING (diagnostic build)
    # DS2I_BUILD_VARIANT=name: a diagnostic / A-B build beside the product (own object directory, library written to
    # profiles/tmp_libs/lib_<name>.so -- git-ignored, travels to the GPU box); the product library is left alone
    variant = os.environ.get("DS2I_BUILD_VARIANT", "")
    lib = LIB
    objdir = os.path.join(CSRC, "build")
    if variant:
        objdir = os.path.join(CSRC, "build", "variant_" + variant)
        libdir = os.path.join(HERE, "..", "profiles", "tmp_libs")
        os.makedirs(libdir, exist_ok=True)
        lib = os.path.join(libdir, "lib_%s.so" % variant)
    else:
        force = force or bool(extra)
    os.makedirs(objdir, exist_ok=True)
    hdrs = _headers()
    objs, jobs = [], []
    for src, name, defs in DEVICE_UNITS + [(h, h, []) for h in HOST_SRCS]:
        sp = os.path.join(CSRC, src)
        obj = os.path.join(objdir, name + ".o")
        objs.append(obj)
        if force or _newer(obj, [sp] + hdrs):
            cmd = [hipcc, "--offload-arch=" + ARCH] + COMMON + extra + defs + ["-c", sp, "-o", obj]
            if src.endswith(".cpp"):
                cmd[1:1] = ["-x", "hip"]
            jobs.append(cmd)
    if jobs:
        from concurrent.futures import ThreadPoolExecutor

        def run(cmd):
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.check_call(cmd)

        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as pool:
            list(pool.map(run, jobs))
    if force or _newer(lib, objs):
        cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", lib] + objs + ["-lpthread"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return lib


if __name__ == "__main__":
    print(build(verbose=True, force="--force" in sys.argv))
