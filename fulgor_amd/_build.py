"""Builds the native pieces in-tree with explicit compiler invocations (hipcc cross-compiles gfx950
without a GPU). Outputs are git-ignored but travel to the GPU box with the working tree."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "fulgor_amd")
CSRC = os.path.join(PKG, "csrc")

LIB_GPU = os.environ.get("FULGOR_LIB_GPU") or os.path.join(PKG, "libfulgor_gpu.so")  # (FULGOR_LIB_GPU: measure another build of the library)
LIB_TOOLS = os.path.join(PKG, "libfgtools.so")
BIN_CCDBG = os.path.join(PKG, "ccdbg_from_fasta")
LIB_ORACLE = os.path.join(ROOT, "oracle", "liboracle.so")


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _walk(d, exts):
    out = []
    for base, _, files in os.walk(d):
        out += [os.path.join(base, f) for f in files if f.endswith(exts)]
    return out


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("build failed: %s\n%s" % (" ".join(cmd), r.stdout))


def build_gpu(force=False):
    srcs = _walk(CSRC, (".hip", ".h", ".hpp")) + [os.path.join(ROOT, "include", "fulgor_gpu.h")]
    if os.environ.get("FULGOR_LIB_GPU"):
        return LIB_GPU
    if force or _newer(LIB_GPU, srcs):
        _run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-pthread",
              os.path.join(CSRC, "fulgor_gpu.hip"), "-o", LIB_GPU, "-lz", "-ldl", "-lhsa-runtime64"])
    return LIB_GPU


def build_tools(force=False):
    src = os.path.join(CSRC, "tools", "readgen.cpp")
    if force or _newer(LIB_TOOLS, [src]):
        _run(["g++", "-O3", "-std=c++17", "-fPIC", "-shared", "-pthread", src, "-o", LIB_TOOLS, "-lz"])
    src = os.path.join(CSRC, "tools", "ccdbg_from_fasta.cpp")
    if force or _newer(BIN_CCDBG, [src]):
        _run(["g++", "-O2", "-std=c++17", src, "-o", BIN_CCDBG, "-lz"])
    return LIB_TOOLS


def build_oracle(force=False):
    d = os.path.join(ROOT, "oracle")
    srcs = [os.path.join(d, f) for f in os.listdir(d) if f.endswith((".cpp", ".hpp"))]
    if force or _newer(LIB_ORACLE, srcs):
        _run(["make", "-C", d, "-B", "liboracle.so"])
    return LIB_ORACLE


def build_all(force=False):
    build_gpu(force)
    build_tools(force)
    build_oracle(force)
    from . import synth
    synth.build_tool()
