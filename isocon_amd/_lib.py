"""ctypes binding of libisocon_hip.so (C ABI: include/isocon_hip.h).

The library is built in-tree by build() (hipcc --offload-arch=gfx950) and loaded from isocon_amd/lib/.  There is
no CPU fallback: if the shared object is missing or no GPU is usable the product functions raise.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("ISOCON_LIB", os.path.join(_HERE, "lib", "libisocon_hip.so"))   # override: kernel experiments
SRC_DIR = os.path.join(_HERE, "csrc")
_SOURCES = sorted(f for f in os.listdir(SRC_DIR) if f.endswith((".hip", ".hpp", ".inc"))) if os.path.isdir(SRC_DIR) else []

ISOCON_OK = 0
ISOCON_E_CAPACITY = -4
NN_INF = 0x3FFFFFFF

u8p = ctypes.POINTER(ctypes.c_uint8)
i8p = ctypes.POINTER(ctypes.c_int8)
u32p = ctypes.POINTER(ctypes.c_uint32)
i32p = ctypes.POINTER(ctypes.c_int32)
u64p = ctypes.POINTER(ctypes.c_uint64)
f32p = ctypes.POINTER(ctypes.c_float)


class NNStats(ctypes.Structure):
    _fields_ = [("pairs_evaluated", ctypes.c_uint64), ("cells_columns", ctypes.c_uint64), ("live_columns", ctypes.c_uint64), ("tiles", ctypes.c_uint64),
                ("hits", ctypes.c_uint64), ("fallback_queries", ctypes.c_uint64), ("full_pairs", ctypes.c_uint64),
                ("kernel_ms", ctypes.c_float), ("scan_kernel_ms", ctypes.c_float), ("seed_kernel_ms", ctypes.c_float),
                ("scan_launches", ctypes.c_uint32), ("pairs_prefiltered", ctypes.c_uint64), ("bound_kernel_ms", ctypes.c_float),
                ("list_kernel_ms", ctypes.c_float), ("lanes_kernel_ms", ctypes.c_float), ("narrow_kernel_ms", ctypes.c_float),
                ("pairs_lanes", ctypes.c_uint64), ("bound_tiles", ctypes.c_uint64), ("pairs_wide_to_lanes", ctypes.c_uint64),
                ("narrow_columns", ctypes.c_uint64), ("pairs_narrow", ctypes.c_uint64), ("pairs_bytes", ctypes.c_uint64),
                ("pairs_block_rejected", ctypes.c_uint64), ("filter_kernel_ms", ctypes.c_float), ("mm_kernel_ms", ctypes.c_float)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


# name -> (restype, argtypes): every symbol include/isocon_hip.h declares
SYMBOLS = {
    "isocon_init": (ctypes.c_int, [ctypes.c_int]),
    "isocon_strerror": (ctypes.c_char_p, [ctypes.c_int]),
    "isocon_last_error": (ctypes.c_char_p, []),
    "isocon_device_count": (ctypes.c_int, []),
    "isocon_release_scratch": (None, []),
    "isocon_store_create": (ctypes.c_int, [u8p, u64p, ctypes.c_uint32, ctypes.POINTER(ctypes.c_void_p)]),
    "isocon_store_create_ptrs": (ctypes.c_int, [u64p, u64p, ctypes.c_uint32, ctypes.POINTER(ctypes.c_void_p)]),
    "isocon_store_create_ptrs_ex": (ctypes.c_int, [u64p, u64p, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(ctypes.c_void_p)]),
    "isocon_host_alloc": (ctypes.c_void_p, [ctypes.c_uint64]),
    "isocon_host_free": (None, [ctypes.c_void_p]),
    "isocon_store_destroy": (None, [ctypes.c_void_p]),
    "isocon_store_size": (ctypes.c_uint32, [ctypes.c_void_p]),
    "isocon_store_device_bytes": (ctypes.c_uint64, [ctypes.c_void_p]),
    "isocon_store_digest": (ctypes.c_int, [ctypes.c_void_p, u64p]),
    "isocon_ed_pairs": (ctypes.c_int, [ctypes.c_void_p, u32p, u32p, i32p, ctypes.c_uint64, i32p, f32p]),
    "isocon_qgram_params": (ctypes.c_int, [i32p]),
    "isocon_qgram_bound_pairs": (ctypes.c_int, [ctypes.c_void_p, u32p, u32p, ctypes.c_uint64, i32p]),
    "isocon_block_bound_pairs": (ctypes.c_int, [ctypes.c_void_p, u32p, u32p, ctypes.c_uint64, ctypes.c_int32, i32p]),
    "isocon_qgram_bound_matrix": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint64, u64p, u8p,
                                                 ctypes.c_uint64, u64p]),
    "isocon_nn_graph": (ctypes.c_int, [ctypes.c_void_p, u8p, u8p, ctypes.c_uint64, i32p, u64p, u32p, ctypes.c_uint64,
                                       u64p, ctypes.POINTER(NNStats)]),
    "isocon_nn_partial": (ctypes.c_int, [ctypes.c_void_p, u8p, u8p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32,
                                         ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int32, i32p, i32p, ctypes.c_uint64, u64p,
                                         ctypes.POINTER(NNStats), u8p]),
    "isocon_msa_build_ops": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint32, u32p, u32p, u64p, u32p, u32p, u32p, u32p, ctypes.c_uint64, u64p, f32p]),
    "isocon_msa_correct_built": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, u32p, u32p, u32p, u8p, ctypes.c_uint32, i32p, u8p,
                                                ctypes.c_uint64, u64p, i32p, ctypes.POINTER(ctypes.c_int64), f32p]),
    "isocon_msa_read_built": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, u8p]),
    "isocon_msa_build_ops_batch": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint32, u32p, u32p, u32p, u64p, u32p, u32p, u32p, u32p, ctypes.c_uint64, u64p, f32p]),
    "isocon_msa_correct_built_batch": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, u32p, u32p, u32p, u8p, ctypes.c_uint32, i32p, u8p,
                                                      ctypes.c_uint64, u64p, i32p, f32p]),
    "isocon_partition_ids": (ctypes.c_int, [ctypes.c_uint32, i32p, ctypes.c_uint64, u32p, u32p, u32p, ctypes.c_int32, u32p, ctypes.POINTER(ctypes.c_int64),
                                            u64p, u32p, u32p]),
    "isocon_nn_finalize": (ctypes.c_int, [ctypes.c_uint32, i32p, i32p, ctypes.c_uint64, i32p, u64p, u32p,
                                          ctypes.c_uint64, u64p]),
    "isocon_nn_partial_dev": (ctypes.c_int, [ctypes.c_void_p, u8p, u8p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32,
                                             ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32, u64p, ctypes.POINTER(NNStats),
                                             u8p]),
    "isocon_nn_hits_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, u64p]),
    "isocon_nn_finalize_dev": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, i32p, u64p, u32p,
                                              ctypes.c_uint64, u64p]),
    "isocon_sg_trace_batch": (ctypes.c_int, [ctypes.c_void_p, u32p, u32p, ctypes.c_uint64, ctypes.c_int32, i8p,
                                             ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, u32p, u64p, ctypes.c_uint64,
                                             u64p, i32p, f32p, i32p]),
    "isocon_sg_strings_batch": (ctypes.c_int, [ctypes.c_void_p, u32p, u32p, ctypes.c_uint64, ctypes.c_int32, i8p,
                                               ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, u32p, u64p, ctypes.c_uint64,
                                               u64p, i32p, u8p, u8p, u64p, ctypes.c_uint64, u64p, f32p, i32p]),
    "isocon_sg_last_stats": (ctypes.c_int, [ctypes.POINTER(ctypes.c_double), ctypes.c_int32]),
    "isocon_exon_filter_from_ops": (ctypes.c_int, [u32p, u64p, ctypes.c_uint64, ctypes.c_int32, ctypes.c_int32, u8p]),
    "isocon_msa_correct": (ctypes.c_int, [u8p, ctypes.c_uint32, ctypes.c_uint32, i32p, u8p, ctypes.c_uint64, u64p, i32p,
                                          ctypes.POINTER(ctypes.c_int64), f32p]),
    "isocon_hw_pairs": (ctypes.c_int, [ctypes.c_void_p, u32p, u32p, i32p, ctypes.c_uint64, i32p, f32p]),
}

_lib = None
_initialised = False


def needs_build() -> bool:
    if not os.path.exists(SO_PATH):
        return True
    t = os.path.getmtime(SO_PATH)
    return any(os.path.getmtime(os.path.join(SRC_DIR, f)) > t for f in _SOURCES if os.path.exists(os.path.join(SRC_DIR, f)))


def build(force: bool = False, verbose: bool = False) -> str:
    """Cross-compile the HIP library for gfx950 (works without a GPU)."""
    if not force and not needs_build():
        build_pyhelp(verbose)
        return SO_PATH
    os.makedirs(os.path.dirname(SO_PATH), exist_ok=True)
    cmd = ["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", "-Wall", "-Wno-unused-function",
           "-o", SO_PATH, os.path.join(SRC_DIR, "isocon_hip.hip")]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=SRC_DIR)
    build_pyhelp(verbose)
    return SO_PATH


def _pyhelp_so():
    # named for THIS interpreter's ABI (sysconfig EXT_SUFFIX): a helper built for another Python is never picked up by mistake
    import sysconfig
    return os.path.join(_HERE, "_pyhelp" + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))


PYHELP_SO = _pyhelp_so()
_pyhelp_warned = False


def _warn_pyhelp(why):
    global _pyhelp_warned
    if not _pyhelp_warned:
        _pyhelp_warned = True
        sys.stderr.write("[isocon_amd] CPython helper unavailable (%s): the wrappers use their pure-Python loops\n" % why)


def build_pyhelp(verbose: bool = False, extra_flags=(), out=None):
    """The CPython helper of the wrappers (cpy/_pyhelp.c: string lists <-> flat buffers); optional -- pure-Python loops otherwise
    (one line on stderr says so).  extra_flags / out: the sanitizer build of tests/test_pyhelp.py."""
    import sysconfig
    src = os.path.join(_HERE, "cpy", "_pyhelp.c")
    out = out or PYHELP_SO
    if not os.path.exists(src):
        return None
    if not extra_flags and os.path.exists(out) and os.path.getmtime(out) >= os.path.getmtime(src):
        return out
    cmd = ["gcc", "-O2", "-fPIC", "-shared", "-pthread", "-Wall", "-I" + sysconfig.get_paths()["include"]] + list(extra_flags) + ["-o", out, src]
    if verbose:
        print(" ".join(cmd))
    try:
        subprocess.check_call(cmd)
    except Exception as e:          # (a read-only install, no compiler)
        _warn_pyhelp("build failed: %s" % e)
        return None
    return out


def pyhelp():
    """the helper module, or None (ISOCON_NO_PYHELP=1 forces None: the tests run the wrappers both ways)"""
    if os.environ.get("ISOCON_NO_PYHELP", "") not in ("", "0"):
        return None
    try:
        from . import _pyhelp
        return _pyhelp
    except Exception as e:
        _warn_pyhelp("import failed: %s" % e)
        return None


def _preload_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm ships its own libamdhip64 (same SONAME as /opt/rocm's, found through its RPATH under
    the un-versioned file name): if this library came first and bound /opt/rocm's copy, a later `import torch` would bring a second
    runtime into the process -- torch then sees no GPU, and device pointers could not be handed from one to the other
    (isocon_amd/dist.py gives RCCL the buffers this library fills).  So torch's copy, when there is one, is loaded first and the
    dynamic linker binds this library's libamdhip64.so.7 to it by SONAME.  ISOCON_HIP_RUNTIME=system keeps /opt/rocm's."""
    if os.environ.get("ISOCON_HIP_RUNTIME", "") == "system":
        return None
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return None
        path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if not os.path.exists(path):
            return None
        return ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
    except Exception:
        return None


def hip_runtimes_loaded():
    """paths of the libamdhip64 copies mapped into this process (more than one: device pointers must not cross libraries)"""
    try:
        with open("/proc/self/maps") as f:
            return sorted({ln.split()[-1] for ln in f if "libamdhip64" in ln})
    except OSError:
        return []


def load():
    """Load the shared object and bind every symbol (no GPU needed for this)."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise RuntimeError("libisocon_hip.so is missing (%s): run `python -c 'import __graft_entry__ as g; g.build()'`; "
                               "there is no CPU fallback" % SO_PATH)
        _preload_torch_hip_runtime()
        L = ctypes.CDLL(SO_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


class IsoconError(RuntimeError):
    pass


def check(rc: int, what: str):
    if rc != ISOCON_OK:
        L = load()
        raise IsoconError("%s failed: %s (%s)" % (what, L.isocon_strerror(rc).decode(), L.isocon_last_error().decode()))


def lib():
    """Library handle with the GPU selected (LOCAL_RANK / ISOCON_GPU_DEVICE / 0).  Raises without a GPU."""
    global _initialised
    L = load()
    if not _initialised:
        dev = int(os.environ.get("ISOCON_GPU_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        check(L.isocon_init(dev), "isocon_init(%d)" % dev)
        _initialised = True
    return L
