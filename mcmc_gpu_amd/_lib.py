"""Build and load libgsm_hip.so (the C-ABI HIP library, include/gsm.h) through ctypes.

There is no CPU fallback: if the library is missing or cannot be loaded, ``load()`` raises and every
product entry point that needs the device fails with it.
"""
from __future__ import annotations

import ctypes as C
import os
import re
import shutil
import subprocess
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
CSRC = PKG_DIR / "csrc"
LIB_PATH = PKG_DIR / "libgsm_hip.so"
HEADER = PKG_DIR.parent / "include" / "gsm.h"
SOURCES = ["gsm_api.hip", "gsm_version.hip", "step_kernel.hip", "step_flux_kernel.hip", "chain_fused_kernel.hip", "chain_strip_kernel.hip", "proposal_kernel.hip", "cholesky_kernel.hip", "sgs_kernel.hip", "pcg64_kernel.hip"]
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
               "-Wno-unused-value", "-Wno-unused-result"]
# per-file extras.  step_flux_kernel: without machine LICM the fp64 polynomial constants of exp() are materialised at
# their use instead of being hoisted out of the step loop into ~30 VGPRs that are then spilled.
# cholesky_kernel: keep the MFMA accumulators in VGPRs -- with 256-thread workgroups LLVM otherwise puts them in AGPRs and
# copies all 32 of them VGPR -> AGPR -> VGPR around the 16 MFMAs of EVERY K step of cz_gemm_kernel (64 v_accvgpr moves and
# a pipeline drain per step: 12.0 -> 9.x ms per 32768 proposals, scripts/kbench.py).
EXTRA_FLAGS = {"step_flux_kernel.hip": ["-mllvm", "-disable-machine-licm"],
               "chain_fused_kernel.hip": ["-mllvm", "-disable-machine-licm"],
               "chain_strip_kernel.hip": ["-mllvm", "-disable-machine-licm"],
               "cholesky_kernel.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}
OBJ_DIR = PKG_DIR / "_build"


class GsmError(RuntimeError):
    """Raised when a libgsm_hip call returns a negative status."""

    def __init__(self, code: int, text: str):
        super().__init__(f"libgsm_hip error {code}: {text}")
        self.code = code


def _diag_flags() -> list[str]:
    """Diagnostic defines taken from the environment (scripts/stamps*.py): part of the build's identity."""
    diag = ["-DGSM_STAMPS"] if os.environ.get("GSM_STAMPS") else []
    if os.environ.get("GSM_STAMP_TID"):
        diag.append("-DGSM_STAMP_TID=" + os.environ["GSM_STAMP_TID"])
    return diag


def source_hash() -> str:
    """First 16 hex digits of the SHA-256 over the library sources (csrc/*.hip, csrc/*.h, include/gsm.h, by name) AND the
    effective compiler flags (a diagnostic build -- GSM_STAMPS -- therefore never passes for the product build)."""
    import hashlib
    h = hashlib.sha256()
    h.update(repr((HIPCC_FLAGS, sorted(EXTRA_FLAGS.items()), _diag_flags())).encode())
    for p in sorted(list(CSRC.glob("*.hip")) + list(CSRC.glob("*.h")), key=lambda q: q.name) + [HEADER]:
        h.update(p.name.encode())
        h.update(p.read_bytes())
    return h.hexdigest()[:16]


def built_hash(path: Path = LIB_PATH):
    """The source hash embedded in a built library (gsm_version's "src:<hash>"), read from the file without loading it."""
    if not path.exists():
        return None
    m = re.search(rb"gsm-hip [0-9.]+ gfx950 src:([0-9a-f]{16})", path.read_bytes())
    return m.group(1).decode() if m else None


def _stale() -> bool:
    return built_hash() != source_hash()


def build(force: bool = False, verbose: bool = False) -> Path:
    """Compile the HIP sources for gfx950 into mcmc_gpu_amd/libgsm_hip.so (in-tree).  Ranks of one node that start together
    (bench.py --gpus N, largeScaleChain_mp) take a file lock: one of them builds, the others find the library current."""
    if not force and not _stale():
        return LIB_PATH
    import fcntl
    OBJ_DIR.mkdir(exist_ok=True)
    with open(OBJ_DIR / ".build.lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not _stale():          # another process built it while this one waited
                return LIB_PATH
            return _build_locked(force, verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(force: bool, verbose: bool) -> Path:
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libgsm_hip.so")
    OBJ_DIR.mkdir(exist_ok=True)
    src_hash = source_hash()
    hdr_t = max(p.stat().st_mtime for p in list(CSRC.glob("*.h")) + [HEADER])
    procs, objs = [], []
    for s in SOURCES:
        src, obj = CSRC / s, OBJ_DIR / (s + ".o")
        objs.append(str(obj))
        is_version = s == "gsm_version.hip"
        if not force and not is_version and obj.exists() and obj.stat().st_mtime > max(src.stat().st_mtime, hdr_t):
            continue
        diag = _diag_flags()
        if is_version:
            diag.append(f'-DGSM_SRC_HASH="{src_hash}"')
        cmd = [hipcc, *HIPCC_FLAGS, *EXTRA_FLAGS.get(s, []), *diag, "-c", "-o", str(obj), str(src)]
        if verbose:
            print(" ".join(cmd))
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for s, pr in procs:
        out, _ = pr.communicate()
        if pr.returncode != 0:
            raise RuntimeError(f"hipcc failed on {s}:\n{out}")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB_PATH), *objs]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc link failed:\n" + r.stdout + r.stderr)
    return LIB_PATH


def declared_symbols() -> list[str]:
    """Function names declared in include/gsm.h."""
    txt = HEADER.read_text()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(gsm_[a-z0-9_]+)\s*\(", txt)))


class RfParams(C.Structure):
    """Mirror of gsm_rf_params (include/gsm.h)."""
    _fields_ = [("range_min_x", C.c_double), ("range_max_x", C.c_double),
                ("range_min_y", C.c_double), ("range_max_y", C.c_double),
                ("scale_min", C.c_double), ("scale_max", C.c_double),
                ("nugget_max", C.c_double), ("smoothness", C.c_double), ("resolution", C.c_double),
                ("model", C.c_int32), ("isotropic", C.c_int32), ("generator", C.c_int32), ("reserved", C.c_int32)]


class SgsBatch(C.Structure):
    """Mirror of gsm_sgs_batch (include/gsm.h)."""
    _fields_ = ([(k, C.c_void_p) for k in ("cur", "next", "proposed", "zcond", "trend", "qt_quantiles", "qt_references", "energy", "state",
                                          "x_axis", "y_axis", "lag_cov", "windows", "cell_off", "cell_cnt", "cells", "z", "cell_base", "u",
                                          "resampled", "loss", "bad", "loss_prev", "accept", "loss_rec", "acc_rec")] +
                [("radius", C.c_double), ("sill", C.c_double), ("cell_off_stride", C.c_int64)] +
                [(k, C.c_int32) for k in ("qt_n", "windowed", "lag_mi", "lag_mj", "hw", "num_points", "max_cells", "use_graph", "grid_finite")])


class Vario(C.Structure):
    """Mirror of gsm_vario (include/gsm.h)."""
    _fields_ = [("azimuth", C.c_double), ("major_range", C.c_double), ("minor_range", C.c_double),
                ("sill", C.c_double), ("nugget", C.c_double), ("s", C.c_double),
                ("vtype", C.c_int32), ("reserved", C.c_int32)]


VTYPE_IDS = {"exponential": 0, "gaussian": 1, "spherical": 2, "matern": 3}


MODEL_IDS = {"Gaussian": 0, "Exponential": 1, "Matern": 2}

_lib = None


def load() -> C.CDLL:
    """Load the library (once) and declare the prototypes of include/gsm.h."""
    global _lib
    if _lib is not None:
        return _lib
    lib_path = Path(os.environ.get("GSM_LIB", LIB_PATH))   # GSM_LIB: another build of the same ABI (A/B measurements)
    if not lib_path.exists():
        raise RuntimeError(
            f"{lib_path} is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  This package has no CPU fallback.")
    # torch ships its own HIP/HSA runtime.  Import it first so that libgsm_hip's NEEDED libamdhip64.so.7
    # resolves to the runtime torch already loaded -- two HIP runtimes in one process cannot both see the GPU.
    import torch  # noqa: F401
    if "GSM_LIB" not in os.environ and built_hash(lib_path) != source_hash():
        # the binary is git-ignored and travels prebuilt: never run kernels that are not the sources' 
        if shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc"):
            build()
        if built_hash(lib_path) != source_hash():
            raise RuntimeError(f"{lib_path} was built from other sources (embedded hash {built_hash(lib_path)}, sources "
                               f"{source_hash()}): rebuild with `python -c 'import __graft_entry__ as g; g.build()'`")
    lib = C.CDLL(str(lib_path))
    vp, i32, i64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double
    lib.gsm_version.restype = C.c_char_p
    lib.gsm_version.argtypes = []
    lib.gsm_last_error.restype = C.c_char_p
    lib.gsm_last_error.argtypes = [vp]
    lib.gsm_create.argtypes = [C.POINTER(vp), i32, i32, i32, i32, i32]
    lib.gsm_destroy.argtypes = [vp]
    lib.gsm_set_static.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, dbl, dbl, vp]
    lib.gsm_set_blocks.argtypes = [vp, i32, C.POINTER(i32), C.POINTER(i32), vp, C.POINTER(i64), vp]
    lib.gsm_set_centres.argtypes = [vp, vp, i32, vp]
    lib.gsm_init_loss.argtypes = [vp, vp, vp, vp, vp, vp]
    lib.gsm_residual.argtypes = [vp, vp, vp, vp]
    lib.gsm_run_replay.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, i64, vp, vp, vp]
    lib.gsm_propose_philox.argtypes = [vp, i32, i64, vp, C.POINTER(RfParams), vp, vp, vp, vp, i64, vp, vp]
    lib.gsm_spectral_from_noise.argtypes = [vp, i32, vp, vp, C.POINTER(RfParams), vp, vp, vp, vp, i64, vp]
    lib.gsm_run_noise.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, C.POINTER(RfParams), vp, vp, vp, i64, vp, vp, vp]
    lib.gsm_run_philox.argtypes = [vp, i32, i64, i32, vp, C.POINTER(RfParams), vp, vp, vp, vp, vp, vp, vp, vp]
    lib.gsm_enable_timing.argtypes = [vp, i32]
    lib.gsm_set_fused.argtypes = [vp, i32]
    lib.gsm_last_run_fused.argtypes = [vp]
    lib.gsm_strip_active.argtypes = [vp]
    lib.gsm_last_timing.argtypes = [vp, C.POINTER(dbl), C.POINTER(i32), C.POINTER(dbl), C.POINTER(i32)]
    lib.gsm_cov_assemble.argtypes = [vp, i32, i32, dbl, C.POINTER(Vario), vp, vp, i64, vp]
    lib.gsm_cholesky_upper.argtypes = [vp, vp, i32, i64, dbl, vp]
    lib.gsm_set_factors.argtypes = [vp, i32, C.POINTER(vp), vp]
    lib.gsm_sgs_blocks.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, dbl, i32, dbl, vp, vp, vp, i32, vp, vp, vp]
    lib.gsm_sgs_loss.argtypes = [vp, vp, vp, vp, vp, vp]
    lib.gsm_sgs_commit.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    lib.gsm_sgs_blocks_batch.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, dbl, i32, dbl, vp, vp, vp, vp, i32, vp]
    lib.gsm_sgs_check.argtypes = [vp, vp]
    lib.gsm_sgs_iterate.argtypes = [vp, C.POINTER(SgsBatch), i32, vp]
    lib.gsm_sgs_graph_replays.argtypes = [vp]
    lib.gsm_sgs_set_kriging.argtypes = [vp, i32, vp]
    lib.gsm_draw_pcg64.argtypes = [vp, i32, C.POINTER(RfParams), vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, vp]
    lib.gsm_sgs_state_init.argtypes = [vp, vp, vp, vp, vp, vp]
    lib.gsm_sgs_finish.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, vp]
    lib.gsm_sgs_draw_philox.argtypes = [vp, vp, i64, i32, vp, vp, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.gsm_sgs_draw_pcg64.argtypes = [vp, vp, i32, vp, vp, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.gsm_qt_transform.argtypes = [vp, vp, vp, i32, vp, vp, i64, i32, vp]
    lib.gsm_sgs_commit_map.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    lib.gsm_sgs_decide.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, i64, vp]
    lib.gsm_min_dist_from_mask.argtypes = [vp, vp, vp, vp, vp, vp]
    lib.gsm_debug_stream_copy.argtypes = [vp, vp, i64, vp]
    lib.gsm_debug_normals.argtypes = [C.c_uint64, i64, C.c_uint32, C.c_uint32, i32, vp, vp]
    lib.gsm_philox_selftest.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    lib.gsm_struct_size.argtypes = [i32]
    for which, cls in ((0, RfParams), (1, SgsBatch), (2, Vario)):       # the ctypes restatements of the header's structs have the library's layout
        if lib.gsm_struct_size(which) != C.sizeof(cls):
            raise RuntimeError(f"{cls.__name__}: ctypes size {C.sizeof(cls)} != library's {lib.gsm_struct_size(which)} (include/gsm.h changed?)")
    for name in declared_symbols():
        fn = getattr(lib, name)   # AttributeError here = header/library mismatch
        if name not in ("gsm_version", "gsm_last_error"):
            fn.restype = C.c_int
    _lib = lib
    return lib


def philox4x32_10(ctr, key):
    """One Philox block through the library (host path; the device uses the same inline function)."""
    lib = load()
    c = (C.c_uint32 * 4)(*[int(v) & 0xFFFFFFFF for v in ctr])
    k = (C.c_uint32 * 2)(*[int(v) & 0xFFFFFFFF for v in key])
    o = (C.c_uint32 * 4)()
    lib.gsm_philox_selftest(c, k, o)
    return [int(v) for v in o]
