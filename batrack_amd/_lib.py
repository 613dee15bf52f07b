"""ctypes binding of include/batrack_ba.h (libbatrack_ba.so, built in-tree by
`batrack_amd.build()` / `__graft_entry__.build()`).  No fallback: if the HIP
library is missing every entry point raises."""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("BT_LIB_PATH") or os.path.join(_HERE, "lib", "libbatrack_ba.so")   # BT_LIB_PATH: measurement builds only
SOURCES = ["ba_kernels.hip", "ba_etile.hip", "ba_stream.hip", "ba_edge2.hip", "ba_edge2u.hip", "ba_dense.hip", "ba_loose.hip", "plan_pack.hip", "plan_device.hip", "ba_plan.cpp", "ba_api.cpp", "se3_kernels.hip", "patchify_kernels.hip", "projective_kernels.hip", "ga_kernels.hip"]
HEADERS = ["ba_kernels.hpp", "ba_plan.hpp", "ba_edge.hpp", "ba_update.hpp", "dev_cache.hpp", "ba_edge2.hpp", "probe.hpp", os.path.join("..", "..", "include", "batrack_ba.h"),
           os.path.join("..", "..", "include", "batrack_se3.h"), os.path.join("..", "..", "include", "batrack_patchify.h"),
           os.path.join("..", "..", "include", "batrack_projective.h"), os.path.join("..", "..", "include", "batrack_ga.h")]
# -fno-slp-vectorize: packed f32 pairs cost more register moves than the packed instructions save (measured on k_edge, round 4's kernel; k_edge2 writes its packed pairs out by hand)
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-munsafe-fp-atomics", "-fno-slp-vectorize"]

BT_OK, BT_EINVAL, BT_ENOMEM, BT_EHIP, BT_EUNSUPPORTED = 0, -1, -2, -3, -4
ERRORS = {BT_EINVAL: "invalid argument", BT_ENOMEM: "out of memory", BT_EHIP: "HIP runtime error",
          BT_EUNSUPPORTED: "unsupported graph (n > 2048 free poses, or a track "
                           "whose edges name more than one source frame: ii must equal ix[kk])"}
LOSS = {"trivial": 0, "huber": 1, "cauchy": 2}


class PlanInfo(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int64) for n in (
        "E", "n_buf", "p_tot", "fixedp", "n_all", "n", "m", "pairs", "tiles", "slots", "erows",
        "max_tile_cams", "nnz_blocks", "updates", "workspace_bytes", "sorted_input")]


class BaArgs(ctypes.Structure):
    _fields_ = [("poses", ctypes.c_void_p), ("patches", ctypes.c_void_p), ("mono_disp", ctypes.c_void_p),
                ("intrinsics", ctypes.c_void_p), ("targets", ctypes.c_void_p), ("target_stride", ctypes.c_int64),
                ("weights", ctypes.c_void_p), ("poses_out", ctypes.c_void_p), ("patches_out", ctypes.c_void_p),
                ("bounds", ctypes.c_float * 4), ("lmbda", ctypes.c_float), ("ep", ctypes.c_float),
                ("alpha", ctypes.c_float), ("loss", ctypes.c_int32), ("structure_only", ctypes.c_int32),
                ("mono_stride", ctypes.c_int64), ("lmbda_per_track", ctypes.c_void_p)]


TORCH_LIB_PATH = os.path.join(_HERE, "lib", "libbatrack_torch.so")
TORCH_SRC = os.path.join(CSRC, "torch_ops.cpp")


class GaArgs(ctypes.Structure):                       # == bt_ga_args in include/batrack_ga.h
    _fields_ = ([(n, ctypes.c_int64) for n in ("T", "N", "S", "gh", "gw", "H", "W", "Q")] +
                [(n, ctypes.c_void_p) for n in ("trajs_2d", "trajs_disp", "trajs_disp_mono", "trajs_vis", "trajs_static", "jj",
                                                "intrinsics", "pose", "query", "trajs_scales", "frame_scales", "frame_shifts")] +
                [("pw_break", ctypes.c_float), ("half_disp", ctypes.c_int32)])


class GaWeights(ctypes.Structure):                    # == bt_ga_weights
    _fields_ = [(n, ctypes.c_float) for n in ("spatial", "rigid", "pts3d", "cam_smooth", "scale_smooth")] + [("smooth_mode", ctypes.c_int32)]


def kernel_sources_sha16():
    """sha256 (16 hex digits) over the sources and headers the HIP library is built from: what a measurement of the kernels
    (profiles/pmc_k_tile.json) is valid for.  bench.py quotes PMC traffic only from a file whose hash equals this."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(SOURCES + HEADERS):
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def _obj_path(src):
    return os.path.join(os.path.dirname(LIB_PATH), "obj", os.path.splitext(src)[0] + ".o")


def build(force=False, verbose=False):
    """hipcc cross-compiles for gfx950 without a GPU present (one object per source, stale ones only, in parallel);
    then the torch.ops registration (host code, g++)."""
    if force or needs_build():
        from concurrent.futures import ThreadPoolExecutor
        os.makedirs(os.path.join(os.path.dirname(LIB_PATH), "obj"), exist_ok=True)
        th = max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS)
        stale = [f for f in SOURCES if force or not os.path.exists(_obj_path(f))
                 or os.path.getmtime(_obj_path(f)) < max(th, os.path.getmtime(os.path.join(CSRC, f)))]
        flags = [f for f in HIPCC_FLAGS if f != "-shared"]

        def compile_one(f):
            cmd = ["hipcc"] + flags + (["-x", "hip"] if f.endswith(".hip") else []) + ["-c", os.path.join(CSRC, f), "-o", _obj_path(f)]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
        with ThreadPoolExecutor(max_workers=min(len(stale), os.cpu_count() or 4) or 1) as ex:
            list(ex.map(compile_one, stale))
        cmd = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + [_obj_path(f) for f in SOURCES] + ["-o", LIB_PATH]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    build_torch_ops(force, verbose)
    return LIB_PATH


def build_torch_ops(force=False, verbose=False):
    """libbatrack_torch.so: TORCH_LIBRARY(batrack_hip) over the C ABI (csrc/torch_ops.cpp)."""
    default_lib = os.path.join(_HERE, "lib", "libbatrack_ba.so")
    if (not force and os.path.exists(TORCH_LIB_PATH) and os.path.getmtime(TORCH_LIB_PATH) >= os.path.getmtime(TORCH_SRC)
            and os.path.getmtime(TORCH_LIB_PATH) >= os.path.getmtime(os.path.join(_HERE, "..", "include", "batrack_ba.h"))):
        return TORCH_LIB_PATH
    import torch
    ti = os.path.join(os.path.dirname(torch.__file__), "include")
    tl = os.path.join(os.path.dirname(torch.__file__), "lib")
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}", "-I" + ti,
           "-I" + os.path.join(ti, "torch", "csrc", "api", "include"), "-I/opt/rocm/include", TORCH_SRC, "-o", TORCH_LIB_PATH,
           "-L" + tl, "-ltorch", "-ltorch_cpu", "-lc10", "-lc10_hip", "-L" + os.path.dirname(default_lib), "-lbatrack_ba",
           "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return TORCH_LIB_PATH


_torch_ops = None
_torch_ops_failed = False


def torch_ops(strict=False):
    """torch.ops.batrack_hip (loads libbatrack_torch.so once), or None when that library is absent or was built against a
    different copy of the C library (BT_LIB_PATH set): the callers then take the ctypes route to the same C ABI — still the
    HIP path, never a CPU fallback.  strict=True raises instead."""
    global _torch_ops, _torch_ops_failed
    if _torch_ops is None and not _torch_ops_failed:
        lib()                                           # the C ABI library first: the registration links against it
        why = None
        if os.environ.get("BT_LIB_PATH"):
            why = "BT_LIB_PATH is set (libbatrack_torch.so links the default libbatrack_ba.so: two copies of the C library would be loaded)"
        elif not os.path.exists(TORCH_LIB_PATH):
            why = f"{TORCH_LIB_PATH} is missing — run `python -c 'import __graft_entry__ as g; g.build()'`"
        else:
            try:
                import torch
                torch.ops.load_library(TORCH_LIB_PATH)
                _torch_ops = torch.ops.batrack_hip
            except OSError as e:
                why = f"{TORCH_LIB_PATH} failed to load: {e}"
        if why is not None:
            _torch_ops_failed = True
            if strict:
                raise RuntimeError("batrack_amd: " + why)
            import warnings
            warnings.warn("batrack_amd: torch.ops.batrack_hip unavailable, using the ctypes binding of the same C ABI: " + why)
    if _torch_ops is None and strict:
        raise RuntimeError("batrack_amd: torch.ops.batrack_hip is unavailable")
    return _torch_ops


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"batrack_amd: HIP library {LIB_PATH} is missing — run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback)")
    # torch bundles its own libamdhip64; it must be in the process BEFORE our library
    # resolves the same SONAME, or the two would talk to different HIP runtimes.
    import torch  # noqa: F401
    L = ctypes.CDLL(LIB_PATH)
    vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
    L.bt_version.restype = i32
    L.bt_plan_jacobian_kernel.restype = i32
    L.bt_plan_jacobian_kernel.argtypes = [vp]
    L.bt_plan_edge_precision.restype = i32
    L.bt_plan_edge_precision.argtypes = [vp]
    L.bt_config_wave_per_tile_kernels.restype = i32
    L.bt_config_wave_per_tile_kernels.argtypes = [i32]
    L.bt_plan_built_on_device.restype = i32
    L.bt_plan_built_on_device.argtypes = [vp]
    L.bt_target_arch.restype = ctypes.c_char_p
    L.bt_plan_create.restype = i32
    L.bt_plan_create.argtypes = [vp, vp, vp, i64, i64, i64, i64, i64, i64, i64, i32, i32, ctypes.POINTER(vp)]
    L.bt_plan_create_shifted.restype = i32
    L.bt_plan_create_shifted.argtypes = [vp, vp, vp, vp, i64, i64, i64, i64, ctypes.POINTER(vp)]
    L.bt_plan_create_shifted_any.restype = i32
    L.bt_plan_create_shifted_any.argtypes = [ctypes.POINTER(vp), i32, vp, vp, vp, i64, i64, i64, i64, ctypes.POINTER(i32), ctypes.POINTER(vp)]
    L.bt_plan_create_shifted_spec.restype = i32
    L.bt_plan_create_shifted_spec.argtypes = [vp, vp, vp, vp, i64, i64, i64, i64, vp, ctypes.POINTER(vp)]
    L.bt_plan_preshift.restype = i32
    L.bt_plan_preshift.argtypes = [vp, i64, ctypes.POINTER(vp)]
    L.bt_plan_spec_bind.restype = i32
    L.bt_plan_spec_bind.argtypes = [vp, vp, vp, vp, i64, i64, i64, i64, vp]
    L.bt_plan_spec_confirm.restype = i32
    L.bt_plan_spec_confirm.argtypes = [vp]
    L.bt_plan_destroy.restype = None
    L.bt_plan_destroy.argtypes = [vp]
    L.bt_plan_pool_trim.restype = None
    L.bt_plan_pool_trim.argtypes = []
    L.bt_plan_get_info.restype = i32
    L.bt_plan_get_info.argtypes = [vp, ctypes.POINTER(PlanInfo)]
    L.bt_plan_workspace_bytes.restype = ctypes.c_size_t
    L.bt_plan_workspace_bytes.argtypes = [vp]
    L.bt_plan_array.restype = i64
    L.bt_plan_array.argtypes = [vp, ctypes.c_char_p, ctypes.POINTER(vp)]
    for name in ("bt_ba_step", "bt_ba_reduce", "bt_ba_solve_update", "bt_ba_pack", "bt_ba_unpack"):
        f = getattr(L, name)
        f.restype = i32
        f.argtypes = [vp, ctypes.POINTER(BaArgs), vp, vp]
    for name in ("bt_ba_reduce_pack", "bt_ba_unpack_solve_update"):
        f = getattr(L, name)
        f.restype = i32
        f.argtypes = [vp, ctypes.POINTER(BaArgs), vp, vp]
    L.bt_xchg_bytes.restype = ctypes.c_size_t
    L.bt_xchg_bytes.argtypes = [vp, i32]
    L.bt_xchg_alloc.restype = i32
    L.bt_xchg_alloc.argtypes = [ctypes.c_size_t, ctypes.POINTER(vp), ctypes.c_char_p]
    L.bt_xchg_open.restype = i32
    L.bt_xchg_open.argtypes = [ctypes.c_char_p, ctypes.POINTER(vp)]
    L.bt_xchg_close.restype = i32
    L.bt_xchg_close.argtypes = [vp]
    L.bt_xchg_free.restype = i32
    L.bt_xchg_free.argtypes = [vp]
    L.bt_ba_reduce_push.restype = i32
    L.bt_ba_reduce_push.argtypes = [vp, ctypes.POINTER(BaArgs), vp, ctypes.POINTER(vp), i32, i32, i64, vp]
    L.bt_ba_pull_solve_update.restype = i32
    L.bt_ba_pull_solve_update.argtypes = [vp, ctypes.POINTER(BaArgs), vp, vp, i32, i64, vp]
    L.bt_ba_xchg_status.restype = i32
    L.bt_ba_xchg_status.argtypes = [vp, vp, vp, ctypes.POINTER(ctypes.c_int32)]
    L.bt_ba_workspace_init.restype = i32
    L.bt_ba_workspace_init.argtypes = [vp, vp, vp]
    L.bt_ba_step_timed.restype = i32
    L.bt_ba_step_timed.argtypes = [vp, ctypes.POINTER(BaArgs), vp, vp, ctypes.POINTER(ctypes.c_float)]
    L.bt_ba_system.restype = vp
    L.bt_ba_system.argtypes = [vp, vp, ctypes.POINTER(i64)]
    L.bt_ba_packed.restype = vp
    L.bt_ba_packed.argtypes = [vp, vp, ctypes.POINTER(i64)]
    L.bt_ba_dx.restype = vp
    L.bt_ba_dx.argtypes = [vp, vp]
    L.bt_ba_status.restype = i32
    L.bt_ba_status.argtypes = [vp, vp, vp, ctypes.POINTER(ctypes.c_int32)]
    for name, nptr in (("bt_se3_exp", 2), ("bt_se3_log", 2), ("bt_se3_inv", 2), ("bt_se3_mul", 3), ("bt_se3_act", 3),
                       ("bt_se3_act4", 3), ("bt_se3_adj", 3), ("bt_se3_adjT", 3), ("bt_se3_matrix", 2)):
        f = getattr(L, name)
        f.restype = i32
        f.argtypes = [vp] * nptr + [i64, i32, vp]
    L.bt_reproject.restype = i32
    L.bt_reproject.argtypes = [vp, i64, vp, i64, i64, vp, vp, vp, vp, i64, i32, vp, vp, vp]
    L.bt_ga_forward.restype = i32
    L.bt_ga_forward.argtypes = [vp, vp, vp, i32, vp]
    L.bt_ga_backward.restype = i32
    L.bt_ga_backward.argtypes = [vp, vp, ctypes.c_float, ctypes.c_float, vp, vp, vp, vp]
    L.bt_ga_backward_total.restype = i32
    L.bt_ga_backward_total.argtypes = [vp, vp, ctypes.POINTER(GaWeights), vp, vp, vp, vp, vp, vp]
    L.bt_ga_mat_to_se3.restype = i32
    L.bt_ga_mat_to_se3.argtypes = [vp, vp, i64, vp]
    L.bt_ga_sample_disp_mono.restype = i32
    L.bt_ga_sample_disp_mono.argtypes = [vp, vp, vp, i64, i64, i64, i64, i64, vp]
    L.bt_patchify.restype = i32
    L.bt_patchify.argtypes = [vp, i64, i64, i64, i64, vp, i64, i32, i32, vp, vp]
    _lib = L
    return L


def check(rc, what):
    if rc != BT_OK:
        raise RuntimeError(f"{what} failed: {ERRORS.get(rc, rc)}")
