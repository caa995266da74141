"""ctypes binding of libfulgor_gpu.so (include/fulgor_gpu.h). Fails loudly when the library is absent:
there is no Python or CPU stand-in for the HIP path."""
import ctypes as C
import importlib.util
import os

from . import _build

_lib = None

u64p = C.POINTER(C.c_uint64)
u32p = C.POINTER(C.c_uint32)


def _share_hip_runtime_with_torch():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so (same SONAME as /opt/rocm's). Two HIP/HSA
    runtimes in one process cannot both own the GPU, so when torch is installed its copy is mapped first
    and libfulgor_gpu.so binds to it; torch later reuses the same instance."""
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec and spec.submodule_search_locations:
        p = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(p):
            C.CDLL(p, mode=C.RTLD_GLOBAL)


def lib():
    global _lib
    if _lib is not None:
        return _lib
    path = _build.LIB_GPU
    if not os.path.exists(path):
        raise RuntimeError(
            "%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). The engine has no fallback path." % path)
    _share_hip_runtime_with_torch()
    L = C.CDLL(path)
    vp = C.c_void_p
    L.fgpu_last_error.restype = C.c_char_p
    L.fgpu_kernel_name.restype = C.c_char_p
    L.fgpu_kernel_name.argtypes = [C.c_int]
    L.fgpu_open.argtypes = [C.c_char_p, C.c_int, C.POINTER(vp)]
    L.fgpu_close.argtypes = [vp]
    L.fgpu_close.restype = None
    L.fgpu_save.argtypes = [vp, C.c_char_p]
    L.fgpu_selfcheck.argtypes = [vp, C.c_uint64]
    L.fgpu_convert.argtypes = [vp, C.c_int, C.c_uint32, C.c_uint32]
    L.fgpu_info.argtypes = [vp, u64p, u64p, u64p, u64p, u64p, C.POINTER(C.c_int)]
    L.fgpu_free.argtypes = [vp]
    L.fgpu_free.restype = None
    for name in ("fgpu_fetch_color_set_ids", "fgpu_full_intersection"):
        getattr(L, name).argtypes = [vp, vp, vp, C.c_uint64, C.POINTER(vp), C.POINTER(vp)]
    L.fgpu_kmer_color_set_ids.argtypes = [vp, vp, vp, C.c_uint64, C.POINTER(vp), C.POINTER(vp)]
    L.fgpu_kmer_matches.argtypes = [vp, vp, vp, C.c_uint64, C.POINTER(vp)]
    L.fgpu_kmer_emitter_create.argtypes = [vp, C.c_int, C.POINTER(vp)]
    L.fgpu_kmer_emitter_add.argtypes = [vp, vp, vp, C.c_uint64, vp, vp, C.POINTER(vp), u64p]
    L.fgpu_kmer_emitter_write.argtypes = [vp, vp, vp, C.c_uint64, vp, vp, C.c_int, u64p]
    L.fgpu_kmer_emitter_free.argtypes = [vp]
    L.fgpu_kmer_emitter_free.restype = None
    L.fgpu_threshold_union.argtypes = [vp, vp, vp, C.c_uint64, C.c_double, C.POINTER(vp), C.POINTER(vp)]
    L.fgpu_intersect_ids.argtypes = [vp, vp, vp, C.c_uint64, C.POINTER(vp), C.POINTER(vp)]
    L.fgpu_reads_upload.argtypes = [vp, vp, vp, C.c_uint64, C.POINTER(vp)]
    L.fgpu_reads_free.argtypes = [vp]
    L.fgpu_reads_free.restype = None
    L.fgpu_result_create.argtypes = [vp, C.POINTER(vp)]
    L.fgpu_result_free.argtypes = [vp]
    L.fgpu_result_free.restype = None
    L.fgpu_run.argtypes = [vp, vp, C.c_uint64, C.c_uint64, C.c_int, C.c_double, vp]
    L.fgpu_run_lookup.argtypes = [vp, vp, C.c_uint64, C.c_uint64, vp]
    L.fgpu_run_colours.argtypes = [vp, C.c_int, C.c_double, vp]
    L.fgpu_result_expand.argtypes = [vp]
    L.fgpu_result_sizes.argtypes = [vp, u64p, u64p, u64p]
    L.fgpu_result_download.argtypes = [vp, vp, vp]
    L.fgpu_result_accumulate_hits.argtypes = [vp, vp, vp]
    L.fgpu_result_format.argtypes = [vp, C.c_int, C.c_uint32, C.POINTER(vp), u64p]
    L.fgpu_result_format_view.argtypes = [vp, C.c_int, C.c_uint32, C.POINTER(vp), u64p]
    L.fgpu_fastx_open.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.fgpu_fastx_open_part.argtypes = [C.c_char_p, C.c_uint, C.c_uint64, C.c_uint64, C.POINTER(vp)]
    L.fgpu_fastx_count.argtypes = [C.c_char_p, C.c_uint, C.c_uint64, C.c_uint64, u64p]
    L.fgpu_fastx_text_size.argtypes = [C.c_char_p, u64p, C.POINTER(C.c_int)]
    L.fgpu_fastx_count_part.argtypes = [vp, u64p]
    L.fgpu_fastx_next.argtypes = [vp, C.c_uint64, C.POINTER(vp), C.POINTER(vp), u64p]
    L.fgpu_fastx_names.argtypes = [vp, C.POINTER(vp), C.POINTER(vp)]
    L.fgpu_fastx_close.argtypes = [vp]
    L.fgpu_fastx_close.restype = None
    L.fgpu_fastx_ring.argtypes = []
    L.fgpu_fastx_ring.restype = C.c_int
    L.fgpu_pseudoalign_stream.argtypes = [vp, vp, C.c_int, C.c_int, C.c_double, C.c_int, C.c_uint64, C.c_int, C.c_uint64, C.c_uint,
                                          u64p, u64p]
    L.fgpu_last_stream_report.argtypes = [C.POINTER(vp)]
    L.fgpu_prepare_host.argtypes = [C.c_int, C.c_uint, C.c_uint, C.c_uint64, C.c_uint64, C.c_int, C.c_uint64, C.c_uint64]
    L.fgpu_stream_prepare.argtypes = [vp, C.c_int, C.c_uint64, C.c_uint, C.c_uint32, C.c_uint64]
    L.fgpu_result_algorithmic_bytes.argtypes = [vp, u64p, u64p, u64p]
    L.fgpu_result_checksum.argtypes = [vp, u64p, u64p]
    L.fgpu_tune.argtypes = [vp, C.c_int, C.c_uint64]
    L.fgpu_result_distinct_lists.argtypes = [vp, u64p]
    L.fgpu_device_report.argtypes = [vp, C.POINTER(vp)]
    L.fgpu_copy_engines_classify.argtypes = [u32p, C.POINTER(C.c_double), C.c_uint32, u32p, u32p]
    L.fgpu_timing_enable.argtypes = [vp, C.c_int]
    L.fgpu_timing_reset.argtypes = [vp]
    L.fgpu_timing_get.argtypes = [vp, C.c_int, C.POINTER(C.c_double), u64p]
    L.fgpu_formatter_create.argtypes = [C.c_int, C.c_uint64, C.POINTER(vp), C.POINTER(vp), u64p]
    L.fgpu_formatter_add.argtypes = [vp, C.c_uint32, vp, vp, C.c_uint64, C.POINTER(vp), u64p]
    L.fgpu_formatter_finish.argtypes = [vp, C.POINTER(vp), u64p]
    L.fgpu_dump.argtypes = [vp, C.c_char_p]
    L.fgpu_export_sizes.argtypes = [vp, u64p, u64p, u64p, u64p, u64p]
    L.fgpu_export.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise RuntimeError("libfulgor_gpu: %s (code %d)" % (lib().fgpu_last_error().decode(), rc))


def take_bytes(ptr, size):
    """copy `size` bytes of a malloc'd library buffer into a bytes object and release the buffer
    (ctypes.string_at takes a C int: outputs of one pass routinely exceed 2 GB)"""
    n = int(size)
    try:
        if n == 0:
            return b""
        out = bytearray(n)
        C.memmove((C.c_char * n).from_buffer(out), ptr.value, n)
        return bytes(out) if n < (1 << 20) else out  # large outputs stay a bytearray: file.write takes either
    finally:
        lib().fgpu_free(ptr)


def copy_array(ptr, count, dtype):
    """numpy copy of `count` items of `dtype` at a library pointer (one memmove; np.ctypeslib.as_array and buffer
    views of ctypes arrays walk large buffers far too slowly)"""
    import numpy as np
    out = np.empty(int(count), dtype=dtype)
    if out.nbytes:
        addr = ptr.value if hasattr(ptr, "value") else C.cast(ptr, C.c_void_p).value
        C.memmove(out.ctypes.data, addr, out.nbytes)
    return out
