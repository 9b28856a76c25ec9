"""ctypes binding of libfhe_ntt.so (include/fhe_ntt.h).

This is plumbing only: every call goes straight to the C ABI, which runs the HIP
kernels.  There is no Python/numpy compute path and no CPU fallback — if the
library is missing, or no GPU is present, calls raise.
"""
import ctypes
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FHE_NTT_LIB") or os.path.join(_HERE, "libfhe_ntt.so")  # env: A/B builds
CSRC = os.path.join(_HERE, "csrc")
SOURCES = ["capi.hip", "ntt_kernels.hip", "ntt_kernels_q62.hip", "ntt_persist.hip", "digit_mac.hip", "digit32.hip", "bfv32.hip", "smallq.hip", "generic63.hip", "zring.hip", "glue.hip"]
HEADERS = ["ntt_kernels.hpp", "ntt_rounds.hpp", "ntt_persist.hpp", "persist_sched.hpp", "digit_mac.hpp", "digit32.hpp", "bfv32.hpp", "smallq.hpp", "ntt32_rounds.hpp", "ntt32_big.hpp", "zq_device.hpp", "capi_internal.hpp", "mac_kernel.hpp", "ntt_kernels.hip",
           os.path.join("..", "..", "include", "fhe_ntt.h"), os.path.join("..", "..", "include", "fhe_ntt_experimental.h")]
OBJ_DIR = os.path.join(_HERE, "build")
# -ffp-contract=off: zring.hip restates the reference's f64 scale-and-round (one IEEE rounding
# per operation); nothing else in the library uses floating point.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result",
               "-ffp-contract=off"]

FHE_OK = 0
FHE_E_BAD_N = -1
FHE_E_BAD_Q = -2
FHE_E_NO_ROOT = -3
FHE_E_NULL = -4
FHE_E_HIP = -5
FHE_E_NO_DEVICE = -6
FHE_E_PARAM_MISMATCH = -7
FHE_E_NOT_CANONICAL = -8
FHE_E_INVALID = -9
# flags of the N3 batch surfaces (include/fhe_ntt.h)
FHE_A_IS_EVALS, FHE_B_IS_EVALS, FHE_OUT_EVALS = 1, 2, 4

# every symbol include/fhe_ntt.h declares (tests check the .so exports them all)
EXPORTS = [
    "fhe_ntt_plan_get", "fhe_ntt_plan_info", "fhe_ntt_plan_tables",
    "fhe_ntt_forward", "fhe_ntt_inverse", "fhe_rq_mul", "fhe_rq_mul_checked",
    "fhe_rq_pointwise_mul", "fhe_rq_check_canonical",
    "fhe_ntt_forward_dev", "fhe_ntt_inverse_dev", "fhe_rq_mul_dev",
    "fhe_rq_mul_workspace_bytes", "fhe_rq_pointwise_mul_dev", "fhe_fill_synthetic_dev",
    "fhe_ntt_set_batch_tile", "fhe_ntt_kernel_timing_enable", "fhe_ntt_kernel_timing_read",
    "fhe_ntt_kernel_timing_reset",
    "fhe_ntt_device_count", "fhe_last_error", "fhe_ntt_version", "fhe_ntt_shutdown",
    "fhe_ntt_plan_prepare", "fhe_ntt_plan_arithmetic", "fhe_ntt_set_check_canonical", "fhe_shard_range", "fhe_shard_gather_dev", "fhe_ntt_release_stream_workspace", "fhe_ntt_workspace_bytes",
    "fhe_tggsw_prepared_words", "fhe_tggsw_prepare_dev", "fhe_tggsw_external_product_prepared_dev",
    "fhe_glwe_ksk_prepared_words", "fhe_glwe_ksk_prepare_dev", "fhe_glwe_key_switch_prepared_dev",
    "fhe_bfv_rlk_prepared_words", "fhe_bfv_rlk_prepare_dev", "fhe_bfv_relinearize_prepared_dev", "fhe_bfv_mul_prepared_dev",
    # next rows: exact products over Z / mod 2^64 on top of the engine
    "fhe_r_naive_mul", "fhe_r_naive_mul_dev", "fhe_mul_div_round_dev",
    "fhe_bfv_tensor", "fhe_bfv_tensor_dev", "fhe_bfv_relinearize_dev", "fhe_bfv_mul", "fhe_bfv_mul_dev",
    "fhe_tn_mul", "fhe_tn_mul_dev", "fhe_tggsw_external_product", "fhe_tggsw_external_product_dev",
    # rows N3/N4: batch surfaces and element-wise glue
    "fhe_tr_dot_dev", "fhe_tr_mul_r_dev", "fhe_glev_mul_dev", "fhe_glwe_key_switch_dev",
    "fhe_tr_dot", "fhe_tr_mul_r", "fhe_glev_mul", "fhe_glwe_key_switch",
    "fhe_tglwe_mul_tn", "fhe_tglwe_mul_tn_dev", "fhe_tglev_mul", "fhe_tglev_mul_dev",
    "fhe_rq_add_dev", "fhe_rq_sub_dev", "fhe_rq_neg_dev", "fhe_rq_mul_by_u64_dev",
    "fhe_rq_mod_switch_dev", "fhe_rq_mul_div_round_dev", "fhe_rq_decompose_dev",
    "fhe_rq_remodule_dev", "fhe_rq_mul_by_f64_dev", "fhe_rq_div_round_dev",
]


# include/fhe_ntt_experimental.h: the persistent kernels' switches (exported, NOT part of the boundary)
EXPORTS_EXPERIMENTAL = ["fhe_ntt_set_persist", "fhe_ntt_persist_status", "fhe_ntt_persist_profile", "fhe_ntt_set_persist_grid"]


class FheError(RuntimeError):
    """A non-zero return of the C ABI — the situations in which the reference panics."""

    def __init__(self, code, msg):
        super().__init__(f"fhe_ntt error {code}: {msg}")
        self.code = code


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def _deps(src):
    return [os.path.join(CSRC, src)] + [os.path.join(CSRC, h) for h in HEADERS]


def needs_build():
    return _stale(LIB_PATH, [d for s in SOURCES for d in _deps(s)])


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950: one object per source (only stale ones are recompiled),
    linked into fhe-study_amd/libfhe_ntt.so (in-tree, so it travels to the GPU box)."""
    if not force and not needs_build():
        return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ_DIR, exist_ok=True)
    objs, procs = [], []
    for src in SOURCES:
        obj = os.path.join(OBJ_DIR, src.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, _deps(src)):
            cmd = [hipcc] + HIPCC_FLAGS + ["-c", "-o", obj, os.path.join(CSRC, src)]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:   # sources compile in parallel (3 processes)
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-o", LIB_PATH] + objs
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB_PATH


_u64 = ctypes.c_uint64
_p64 = ctypes.POINTER(ctypes.c_uint64)
_vp = ctypes.c_void_p
_sz = ctypes.c_size_t
_int = ctypes.c_int

_lib = None


def _preload_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so (same SONAME as /opt/rocm's).
    Two HIP runtimes in one process cannot both own the GPU, so if torch is installed its
    copy is loaded first and libfhe_ntt.so binds to it — whichever of the two the process
    imports first.  torch itself is NOT imported here."""
    if "torch" in sys.modules:
        return
    try:
        import importlib.util

        spec = importlib.util.find_spec("torch")
        if spec and spec.origin:
            cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
            if os.path.exists(cand):
                ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
    except Exception:
        pass  # fall back to the loader's default search (RUNPATH /opt/rocm/lib)


def load_library():
    """dlopen the in-tree library; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError(
            f"{LIB_PATH} is missing — run `python -c 'import __graft_entry__ as g; g.build()'`; "
            "there is no fallback path")
    _preload_hip_runtime()
    L = ctypes.CDLL(LIB_PATH)
    L.fhe_ntt_version.restype = ctypes.c_char_p
    if b"ABLATED" in L.fhe_ntt_version() and os.environ.get("FHE_NTT_ALLOW_ABLATED") != "1":
        # a timing-only build (tools/abl_build.sh: kernels that skip work and return wrong words by design), reached
        # through a stray FHE_NTT_LIB or -D flag: never run tests or benches on it by accident
        raise RuntimeError(f"{LIB_PATH} reports {L.fhe_ntt_version().decode()!r}: a timing-only build with wrong results "
                           "by design; set FHE_NTT_ALLOW_ABLATED=1 to load it on purpose (tools/abl_*.py do)")
    L.fhe_ntt_plan_get.argtypes = [_u64, _u64, ctypes.POINTER(_vp)]
    L.fhe_ntt_plan_info.argtypes = [_vp, _p64, _p64, _p64, _p64]
    L.fhe_ntt_plan_tables.argtypes = [_vp, _p64, _p64]
    L.fhe_ntt_forward.argtypes = [_vp, _vp, _vp, _sz]
    L.fhe_ntt_inverse.argtypes = [_vp, _vp, _vp, _sz]
    L.fhe_rq_mul.argtypes = [_vp, _vp, _int, _vp, _int, _vp, _vp, _vp, _vp, _sz]
    L.fhe_rq_mul_checked.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, _sz]
    L.fhe_rq_pointwise_mul.argtypes = [_vp, _vp, _vp, _vp, _sz]
    L.fhe_rq_check_canonical.argtypes = [_vp, _vp, _sz]
    L.fhe_ntt_forward_dev.argtypes = [_vp, _vp, _vp, _sz, _vp]
    L.fhe_ntt_inverse_dev.argtypes = [_vp, _vp, _vp, _sz, _vp]
    L.fhe_rq_mul_dev.argtypes = [_vp, _vp, _int, _vp, _int, _vp, _vp, _vp, _vp, _sz, _vp, _vp]
    L.fhe_rq_mul_workspace_bytes.argtypes = [_vp, _sz]
    L.fhe_rq_mul_workspace_bytes.restype = _sz
    L.fhe_rq_pointwise_mul_dev.argtypes = [_vp, _vp, _vp, _vp, _sz, _vp]
    L.fhe_fill_synthetic_dev.argtypes = [_u64, _u64, _u64, _sz, _vp, _vp]
    L.fhe_ntt_set_batch_tile.argtypes = [_sz]
    L.fhe_ntt_set_persist.argtypes = [ctypes.c_uint, ctypes.c_uint, ctypes.c_uint, ctypes.c_uint]
    L.fhe_ntt_persist_status.argtypes = []
    L.fhe_ntt_persist_profile.argtypes = [_vp]
    L.fhe_ntt_set_persist_grid.argtypes = [_int]
    L.fhe_ntt_kernel_timing_enable.argtypes = [_int]
    L.fhe_ntt_kernel_timing_read.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_double), _p64, _int]
    L.fhe_ntt_kernel_timing_reset.argtypes = []
    _i64p = ctypes.POINTER(ctypes.c_int64)
    _uint = ctypes.c_uint
    L.fhe_r_naive_mul.argtypes = [_u64, _vp, _vp, _vp, _sz]
    L.fhe_r_naive_mul_dev.argtypes = [_u64, _vp, _vp, _vp, _sz, _uint, _uint, _vp]
    L.fhe_mul_div_round_dev.argtypes = [_u64, _u64, _vp, _u64, _u64, _vp, _sz, _vp]
    L.fhe_bfv_tensor.argtypes = [_u64, _u64, _u64, _vp, _vp, _sz]
    L.fhe_bfv_tensor_dev.argtypes = [_u64, _u64, _u64, _vp, _vp, _sz, _vp]
    L.fhe_bfv_relinearize_dev.argtypes = [_u64, _u64, _u64, _vp, _vp, _vp, _sz, _vp]
    L.fhe_bfv_mul.argtypes = [_u64, _u64, _u64, _u64, _vp, _vp, _vp, _sz]
    L.fhe_bfv_mul_dev.argtypes = [_u64, _u64, _u64, _u64, _vp, _vp, _vp, _sz, _vp]
    L.fhe_tn_mul.argtypes = [_u64, _vp, _vp, _vp, _sz]
    L.fhe_tn_mul_dev.argtypes = [_u64, _vp, _vp, _vp, _sz, _vp]
    L.fhe_tggsw_external_product.argtypes = [_u64, _uint, _uint, _vp, _vp, _vp, _sz]
    L.fhe_tggsw_external_product_dev.argtypes = [_u64, _uint, _uint, _vp, _vp, _vp, _sz, _vp]
    L.fhe_bfv_rlk_prepared_words.argtypes = [_u64, _u64, _u64]
    L.fhe_bfv_rlk_prepared_words.restype = _sz
    L.fhe_bfv_rlk_prepare_dev.argtypes = [_u64, _u64, _u64, _vp, _vp, _vp]
    L.fhe_bfv_relinearize_prepared_dev.argtypes = [_u64, _u64, _u64, _vp, _vp, _vp, _sz, _vp]
    L.fhe_bfv_mul_prepared_dev.argtypes = [_u64, _u64, _u64, _u64, _vp, _vp, _vp, _sz, _vp]
    L.fhe_tggsw_prepared_words.argtypes = [_u64, _uint, _uint]
    L.fhe_tggsw_prepared_words.restype = _sz
    L.fhe_tggsw_prepare_dev.argtypes = [_u64, _uint, _uint, _vp, _vp, _vp]
    L.fhe_tggsw_external_product_prepared_dev.argtypes = [_u64, _uint, _uint, _vp, _vp, _vp, _sz, _vp]
    L.fhe_glwe_ksk_prepared_words.argtypes = [_vp, _uint, _uint, _uint]
    L.fhe_glwe_ksk_prepared_words.restype = _sz
    L.fhe_glwe_ksk_prepare_dev.argtypes = [_vp, _uint, _uint, _uint, _vp, _vp, _vp]
    L.fhe_glwe_key_switch_prepared_dev.argtypes = [_vp, _uint, _uint, _uint, _vp, _vp, _vp, _sz, _vp]
    L.fhe_tr_dot_dev.argtypes = [_vp, _vp, _vp, _vp, _uint, _sz, _uint, _vp]
    L.fhe_tr_mul_r_dev.argtypes = [_vp, _vp, _vp, _vp, _uint, _sz, _uint, _vp]
    L.fhe_glev_mul_dev.argtypes = [_vp, _uint, _uint, _vp, _vp, _vp, _sz, _uint, _vp]
    L.fhe_glwe_key_switch_dev.argtypes = [_vp, _uint, _uint, _uint, _vp, _vp, _vp, _sz, _uint, _vp]
    L.fhe_tglwe_mul_tn.argtypes = [_u64, _uint, _vp, _vp, _vp, _sz]
    L.fhe_tglwe_mul_tn_dev.argtypes = [_u64, _uint, _vp, _vp, _vp, _sz, _vp]
    L.fhe_tglev_mul.argtypes = [_u64, _uint, _uint, _vp, _vp, _vp, _sz]
    L.fhe_tglev_mul_dev.argtypes = [_u64, _uint, _uint, _vp, _vp, _vp, _sz, _vp]
    L.fhe_tr_dot.argtypes = [_vp, _vp, _vp, _vp, _uint, _sz]
    L.fhe_tr_mul_r.argtypes = [_vp, _vp, _vp, _vp, _uint, _sz]
    L.fhe_glev_mul.argtypes = [_vp, _uint, _uint, _vp, _vp, _vp, _sz]
    L.fhe_glwe_key_switch.argtypes = [_vp, _uint, _uint, _uint, _vp, _vp, _vp, _sz]
    L.fhe_rq_add_dev.argtypes = [_vp, _vp, _vp, _vp, _sz, _vp]
    L.fhe_rq_sub_dev.argtypes = [_vp, _vp, _vp, _vp, _sz, _vp]
    L.fhe_rq_neg_dev.argtypes = [_vp, _vp, _vp, _sz, _vp]
    L.fhe_rq_mul_by_u64_dev.argtypes = [_vp, _vp, _u64, _vp, _sz, _vp]
    L.fhe_rq_mod_switch_dev.argtypes = [_u64, _u64, _vp, _vp, _sz, _vp]
    L.fhe_rq_mul_div_round_dev.argtypes = [_u64, _u64, _u64, _vp, _vp, _sz, _vp]
    L.fhe_rq_decompose_dev.argtypes = [_u64, _u64, _uint, _uint, _vp, _vp, _sz, _vp]
    L.fhe_rq_remodule_dev.argtypes = [_u64, _vp, _vp, _sz, _vp]
    L.fhe_rq_mul_by_f64_dev.argtypes = [_u64, ctypes.c_double, _vp, _vp, _sz, _vp]
    L.fhe_rq_div_round_dev.argtypes = [_u64, _u64, _vp, _vp, _sz, _vp]
    L.fhe_ntt_device_count.argtypes = []
    L.fhe_ntt_plan_prepare.argtypes = [_vp]
    L.fhe_ntt_plan_arithmetic.argtypes = [_vp]
    L.fhe_ntt_workspace_bytes.argtypes = []
    L.fhe_ntt_workspace_bytes.restype = _sz
    L.fhe_ntt_release_stream_workspace.argtypes = [_vp]
    L.fhe_ntt_set_check_canonical.argtypes = [_int]
    L.fhe_shard_range.argtypes = [_sz, _uint, _uint, ctypes.POINTER(_sz), ctypes.POINTER(_sz)]
    L.fhe_shard_gather_dev.argtypes = [_sz, _sz, _uint, ctypes.POINTER(_int), ctypes.POINTER(_vp), _int, _vp, _vp]
    L.fhe_last_error.restype = ctypes.c_char_p
    L.fhe_ntt_version.restype = ctypes.c_char_p
    for name in EXPORTS:
        fn = getattr(L, name)
        if fn.restype is ctypes.c_int and name not in ("fhe_last_error", "fhe_ntt_version"):
            fn.restype = ctypes.c_int
    _lib = L
    return L


def _check(rc):
    if rc != FHE_OK:
        raise FheError(rc, load_library().fhe_last_error().decode())


def _host(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return a, a.ctypes.data_as(_vp)


class Plan:
    """(q,n)-keyed plan — the reference's `roots(q,n)` cache entry, arith/src/ntt.rs:18-38."""

    def __init__(self, q, n):
        L = load_library()
        h = _vp()
        _check(L.fhe_ntt_plan_get(int(q), int(n), ctypes.byref(h)))
        self.handle = h
        self.q, self.n = int(q), int(n)

    ARITH_NAMES = {0: "shoup62", 1: "shoup61", 2: "pseudo-mersenne", 3: "word32", 4: "strict63", 5: "montgomery"}

    def arithmetic(self):
        """fhe_ntt_plan_arithmetic: which exact form of Zq::mul the transform kernels run for this modulus"""
        rc = load_library().fhe_ntt_plan_arithmetic(self.handle)
        if rc < 0:
            _check(rc)
        return rc

    def info(self):
        q, n, psi, ninv = _u64(), _u64(), _u64(), _u64()
        _check(load_library().fhe_ntt_plan_info(self.handle, q, n, psi, ninv))
        return dict(q=q.value, n=n.value, psi=psi.value, n_inv=ninv.value)

    def tables(self):
        r = np.empty(self.n, dtype=np.uint64)
        ri = np.empty(self.n, dtype=np.uint64)
        _check(load_library().fhe_ntt_plan_tables(self.handle, r.ctypes.data_as(_p64),
                                                  ri.ctypes.data_as(_p64)))
        return r, ri

    # -- host buffers ---------------------------------------------------------
    def _batch(self, a):
        if a.size % self.n:
            raise ValueError(f"array of {a.size} values is not a whole number of n={self.n} polynomials")
        return a.size // self.n

    def forward(self, a):
        a, pa = _host(a)
        out = np.empty_like(a)
        _check(load_library().fhe_ntt_forward(self.handle, pa, out.ctypes.data_as(_vp), self._batch(a)))
        return out

    def inverse(self, a):
        a, pa = _host(a)
        out = np.empty_like(a)
        _check(load_library().fhe_ntt_inverse(self.handle, pa, out.ctypes.data_as(_vp), self._batch(a)))
        return out

    def rq_mul(self, a, b, a_is_evals=False, b_is_evals=False, want_evals=True):
        """→ (c, c_evals, a_evals, b_evals); the last three are None unless want_evals."""
        a, pa = _host(a)
        b, pb = _host(b)
        if a.shape != b.shape:
            raise ValueError("operand shapes differ")
        c = np.empty_like(a)
        ce = np.empty_like(a) if want_evals else None
        ae = np.empty_like(a) if want_evals else None
        be = np.empty_like(a) if want_evals else None
        p = lambda x: x.ctypes.data_as(_vp) if x is not None else None
        _check(load_library().fhe_rq_mul(self.handle, pa, int(a_is_evals), pb, int(b_is_evals),
                                         p(c), p(ce), p(ae), p(be), self._batch(a)))
        return c, ce, ae, be

    def pointwise_mul(self, a, b):
        a, pa = _host(a)
        b, pb = _host(b)
        c = np.empty_like(a)
        _check(load_library().fhe_rq_pointwise_mul(self.handle, pa, pb, c.ctypes.data_as(_vp),
                                                   self._batch(a)))
        return c

    def check_canonical(self, a):
        a, pa = _host(a)
        _check(load_library().fhe_rq_check_canonical(self.handle, pa, self._batch(a)))

    # -- device pointers (ints), stream handle (int or None) -----------------------
    def forward_dev(self, d_in, d_out, batch, stream=None):
        _check(load_library().fhe_ntt_forward_dev(self.handle, d_in, d_out, batch, stream))

    def inverse_dev(self, d_in, d_out, batch, stream=None):
        _check(load_library().fhe_ntt_inverse_dev(self.handle, d_in, d_out, batch, stream))

    def rq_mul_dev(self, d_a, d_b, d_c, batch, a_is_evals=False, b_is_evals=False, d_c_evals=None,
                   d_a_evals=None, d_b_evals=None, d_work=None, stream=None):
        _check(load_library().fhe_rq_mul_dev(self.handle, d_a, int(a_is_evals), d_b, int(b_is_evals),
                                             d_c, d_c_evals, d_a_evals, d_b_evals, batch, d_work, stream))

    def pointwise_mul_dev(self, d_a, d_b, d_c, batch, stream=None):
        _check(load_library().fhe_rq_pointwise_mul_dev(self.handle, d_a, d_b, d_c, batch, stream))

    def workspace_bytes(self, batch):
        return int(load_library().fhe_rq_mul_workspace_bytes(self.handle, batch))


def rq_mul_checked(plan_a, plan_b, a, b):
    a, pa = _host(a)
    b, pb = _host(b)
    c = np.empty_like(a)
    _check(load_library().fhe_rq_mul_checked(plan_a.handle, plan_b.handle, pa, pb,
                                             c.ctypes.data_as(_vp), None, a.size // plan_a.n))
    return c


# ---- next rows: exact products over Z / mod 2^64 (host buffers) -----------------------------

def r_naive_mul(n, a, b):
    """arith::ring_n::naive_mul (ring_n.rs:307-320) → (batch, 2n-1) int64"""
    a = np.ascontiguousarray(a, dtype=np.int64)
    b = np.ascontiguousarray(b, dtype=np.int64)
    batch = a.size // n
    out = np.empty(batch * 2 * n, dtype=np.int64)
    _check(load_library().fhe_r_naive_mul(n, a.ctypes.data_as(_vp), b.ctypes.data_as(_vp),
                                          out.ctypes.data_as(_vp), batch))
    return out.reshape(batch, 2 * n)[:, :2 * n - 1]


def bfv_tensor(q, n, t, a0, a1, b0, b1):
    """RLWE::tensor (bfv/src/lib.rs:59-85) → (c0, c1, c2), each (batch, n)"""
    ab = np.ascontiguousarray(np.stack([np.asarray(x, dtype=np.uint64).reshape(-1, n) for x in (a0, a1, b0, b1)]))
    batch = ab.shape[1]
    c = np.empty((3, batch, n), dtype=np.uint64)
    _check(load_library().fhe_bfv_tensor(q, n, t, ab.ctypes.data_as(_vp), c.ctypes.data_as(_vp), batch))
    return c[0], c[1], c[2]


def bfv_mul(q, n, t, pq, rlk0, rlk1, a0, a1, b0, b1):
    """RLWE::mul (bfv/src/lib.rs:87-90) → (o0, o1), each (batch, n)"""
    ab = np.ascontiguousarray(np.stack([np.asarray(x, dtype=np.uint64).reshape(-1, n) for x in (a0, a1, b0, b1)]))
    rlk = np.ascontiguousarray(np.stack([np.asarray(rlk0, dtype=np.uint64), np.asarray(rlk1, dtype=np.uint64)]))
    batch = ab.shape[1]
    out = np.empty((2, batch, n), dtype=np.uint64)
    _check(load_library().fhe_bfv_mul(q, n, t, pq, rlk.ctypes.data_as(_vp), ab.ctypes.data_as(_vp),
                                      out.ctypes.data_as(_vp), batch))
    return out[0], out[1]


def tn_mul(n, a, b):
    """Tn x Tn (ring_torus.rs:266-298) → (batch, n) uint64"""
    a, pa = _host(a)
    b, pb = _host(b)
    out = np.empty_like(a)
    _check(load_library().fhe_tn_mul(n, pa, pb, out.ctypes.data_as(_vp), a.size // n))
    return out


def tglwe_mul_tn(n, k, tglwe, p):
    """TGLWE x Tn (tfhe/src/tglwe.rs:182-194); tglwe [batch][(k+1)][n], p [batch][n]"""
    c, pc = _host(tglwe)
    t, pt = _host(p)
    out = np.empty_like(c)
    _check(load_library().fhe_tglwe_mul_tn(n, k, pc, pt, out.ctypes.data_as(_vp), t.size // n))
    return out


def tglev_mul(n, k, l, tglev, v):
    """TGLev x Vec<Tn> (tfhe/src/tggsw.rs:139-149); tglev [l][(k+1)][n], v [batch][l][n] -> [batch][(k+1)][n]"""
    g, pg = _host(tglev)
    d, pd = _host(v)
    batch = d.size // (l * n)
    out = np.empty(batch * (k + 1) * n, dtype=np.uint64)
    _check(load_library().fhe_tglev_mul(n, k, l, pg, pd, out.ctypes.data_as(_vp), batch))
    return out.reshape(batch, k + 1, n)


def tggsw_external_product(n, k, l, tggsw, tglwe):
    """TGGSW x TGLWE (tfhe/src/tggsw.rs:45-62); tggsw [(k+1)][l][(k+1)][n], tglwe [batch][(k+1)][n]"""
    g, pg = _host(tggsw)
    t, pt = _host(tglwe)
    batch = t.size // ((k + 1) * n)
    out = np.empty_like(t)
    _check(load_library().fhe_tggsw_external_product(n, k, l, pg, pt, out.ctypes.data_as(_vp), batch))
    return out


def fill_synthetic_dev(q, seed, first_index, count, d_out, stream=None):
    _check(load_library().fhe_fill_synthetic_dev(int(q), int(seed), int(first_index), int(count),
                                                 d_out, stream))


def shard_gather_dev(total_rows, row_words, src_devices, d_src_shards, dst_device, d_dst, stream=None):
    """fhe_shard_gather_dev: the shards of a block-partitioned batch (entry r = device pointer to rank r's rows of
    shard_range(total_rows, world, r), or None for an empty shard) copied in rank order onto dst_device — no torch, no RCCL."""
    world = len(src_devices)
    devs = (_int * world)(*[int(d) for d in src_devices])
    ptrs = (_vp * world)(*[(_vp(int(p)) if p else _vp(None)) for p in d_src_shards])
    _check(load_library().fhe_shard_gather_dev(int(total_rows), int(row_words), world, devs, ptrs, int(dst_device), d_dst, stream))


def device_count():
    return int(load_library().fhe_ntt_device_count())


def set_check_canonical(on):
    """FHE_NTT_CHECK_CANONICAL at run time: transforms reject inputs >= q instead of returning garbage."""
    _check(load_library().fhe_ntt_set_check_canonical(int(bool(on))))


def shard_range(total, world, rank):
    """[begin, end) of `total` units owned by `rank` of `world` (fhe_shard_range; host-only)."""
    b, e = _sz(0), _sz(0)
    _check(load_library().fhe_shard_range(int(total), int(world), int(rank), ctypes.byref(b), ctypes.byref(e)))
    return int(b.value), int(e.value)


def set_batch_tile(polys):
    _check(load_library().fhe_ntt_set_batch_tile(int(polys)))


def set_persist(mode, tile_polys=1, lag=1, ringslots=4):
    """The one-launch n = 2^16 forward transform (csrc/ntt_persist.hip): mode 0 off (two-pass kernels), "A" / 1 lagged
    tiles (tile_polys, lag, ringslots; ringslots 0 = through the output buffer), "B" / 2 teams (ringslots)."""
    mode = {"A": 1, "B": 2, "D": 3, "E": 4, "a": 1, "b": 2, "d": 3, "e": 4}.get(mode, mode)
    _check(load_library().fhe_ntt_set_persist(int(mode), int(tile_polys), int(lag), int(ringslots)))


def set_persist_grid(workgroups):
    """Workgroups of a persistent launch (0: as many as the chip holds)."""
    _check(load_library().fhe_ntt_set_persist_grid(int(workgroups)))


def persist_status():
    """Raises FheError if a persistent launch that has finished gave up a bounded wait (call after synchronising)."""
    _check(load_library().fhe_ntt_persist_status())


def kernel_timing_enable(on):
    _check(load_library().fhe_ntt_kernel_timing_enable(int(bool(on))))


def kernel_timing_reset():
    _check(load_library().fhe_ntt_kernel_timing_reset())


def kernel_timing_read(cap=64):
    """→ {kernel_name: (total_ms, launches)}"""
    names = ctypes.create_string_buffer(64 * cap)
    ms = (ctypes.c_double * cap)()
    cnt = (ctypes.c_uint64 * cap)()
    k = load_library().fhe_ntt_kernel_timing_read(names, ms, cnt, cap)
    out = {}
    for i in range(min(k, cap)):
        nm = names.raw[i * 64:(i + 1) * 64].split(b"\0", 1)[0].decode()
        out[nm] = (float(ms[i]), int(cnt[i]))
    return out


def shutdown():
    global _lib
    if _lib is not None:
        _lib.fhe_ntt_shutdown()
