"""CPU: the drop-in boundary — libfhe_ntt.so loads, exports every symbol that
include/fhe_ntt.h declares, builds plans on the host exactly like the reference's
`roots(q,n)`, reports the reference's panics as error codes, and refuses to
compute without a GPU (no CPU fallback).  No compute calls here."""
import os
import re

import numpy as np
import pytest

from conftest import Q16, Q61, ROOT, golden_cases, load_golden


def declared_symbols(header="fhe_ntt.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fhe_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(pkg):
    lib = pkg.load_library()
    syms = declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/fhe_ntt.h but not exported"
    assert sorted(pkg.binding.EXPORTS) == syms
    # the persistent kernels' switches are exported too, but declared OUTSIDE the boundary header (round 5)
    exp = declared_symbols("fhe_ntt_experimental.h")
    assert exp == sorted(pkg.binding.EXPORTS_EXPERIMENTAL) and not set(exp) & set(syms)
    for s in exp:
        assert hasattr(lib, s), f"{s} declared in include/fhe_ntt_experimental.h but not exported"


def test_product_does_not_touch_the_oracle():
    # the product tree must never import/link/execute anything under oracle/
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "fhe-study_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h", ".c")) or f == "Makefile":
                p = os.path.join(dirpath, f)
                s = open(p, errors="ignore").read()
                if re.search(r"(from|import)\s+oracle|liboracle|oracle/", s) and "tests" not in p:
                    bad.append(p)
    assert not bad, bad


@pytest.mark.parametrize("name", golden_cases())
def test_plan_tables_match_golden(pkg, name):
    g = load_golden(name)
    plan = pkg.Plan(int(g["q"]), int(g["n"]))
    info = plan.info()
    assert info["psi"] == int(g["psi"]) and info["n_inv"] == int(g["n_inv"])
    r, ri = plan.tables()
    assert np.array_equal(r, g["roots"]) and np.array_equal(ri, g["roots_inv"])


def test_plan_tables_match_oracle_large(pkg, oracle):
    for q, n in ((Q61, 8192), (Q61, 65536), (Q16, 32768), (Q61, 1 << 17)):
        r, ri = pkg.Plan(q, n).tables()
        orr, ori, n_inv, psi = oracle.roots(q, n)
        assert np.array_equal(r, orr) and np.array_equal(ri, ori)
        assert pkg.Plan(q, n).info() == dict(q=q, n=n, psi=psi, n_inv=n_inv)


def test_plan_is_memoised(pkg):
    assert pkg.Plan(Q61, 1024).handle.value == pkg.Plan(Q61, 1024).handle.value


def test_error_codes_mirror_reference_panics(pkg):
    B = pkg.binding
    cases = [
        (Q16, 3, B.FHE_E_BAD_N),         # assert!(n.is_power_of_two()), ntt.rs:116
        (Q16, 0, B.FHE_E_BAD_N),
        (Q16, 1, B.FHE_E_BAD_N),         # degenerate shift, ntt.rs:139
        (Q16, 1 << 16, B.FHE_E_BAD_Q),   # assert!((q-1) % n == 0), ntt.rs:117 (2n = 2^17 does not divide 2^16)
        (7, 4, B.FHE_E_BAD_Q),
        (Q61, 1 << 21, B.FHE_E_BAD_N),   # engine limit
        ((1 << 63) + 9, 4, B.FHE_E_BAD_Q),   # q >= 2^63: the reference's Zq::add overflows there (zq.rs:225)
        (2, 2, B.FHE_E_BAD_Q),
    ]
    for q, n, code in cases:
        with pytest.raises(pkg.FheError) as ei:
            pkg.Plan(q, n)
        assert ei.value.code == code, (q, n, ei.value)


def test_composite_modulus_is_not_detected_like_the_reference(pkg, oracle):
    # The reference assumes q prime (ring_nq.rs:17) and never checks: for q = 289 = 17^2,
    # n = 16 the k-search (ntt.rs:120-129) returns a w with w^16 != 1 that is not a root of
    # unity at all, and the Fermat "inverses" are not inverses.  The plan mirrors that
    # behaviour value for value instead of inventing a check the reference lacks.
    plan = pkg.Plan(289, 16)
    r, ri = plan.tables()
    orr, ori, n_inv, psi = oracle.roots(289, 16)
    assert plan.info()["psi"] == psi and plan.info()["n_inv"] == n_inv
    assert np.array_equal(r, orr) and np.array_equal(ri, ori)


def test_param_mismatch_is_reported_before_any_compute(pkg):
    a = np.zeros(4, dtype=np.uint64)
    with pytest.raises(pkg.FheError) as ei:
        pkg.binding.rq_mul_checked(pkg.Plan(Q16, 4), pkg.Plan(Q16, 8), a, a)
    assert ei.value.code == pkg.binding.FHE_E_PARAM_MISMATCH
    x = pkg.Rq.from_vec_u64(pkg.RingParam(Q16, 4), [1, 2, 3, 4])
    y = pkg.Rq.from_vec_u64(pkg.RingParam(Q16, 8), list(range(8)))
    with pytest.raises(pkg.FheError):
        x * y


def test_null_and_empty_arguments(pkg):
    lib = pkg.load_library()
    assert lib.fhe_ntt_forward(None, None, None, 1) == pkg.binding.FHE_E_NULL
    plan = pkg.Plan(Q16, 4)
    assert lib.fhe_ntt_forward(plan.handle, None, None, 0) == 0   # empty batch is a no-op
    assert lib.fhe_ntt_forward(plan.handle, None, None, 1) == pkg.binding.FHE_E_NULL
    assert lib.fhe_ntt_plan_get(Q16, 4, None) == pkg.binding.FHE_E_NULL
    assert b"NULL" in lib.fhe_last_error()


def test_from_vec_u64_folds_like_the_reference(pkg):
    # arith/src/ring_nq.rs:626-650 test_polynomial_ring (q=7): X^n+1 fold + mod q
    p = pkg.Rq.from_vec_u64(pkg.RingParam(7, 4), [0, 1, 2, 3, 4, 5])
    assert p.coeffs.tolist() == [3, 3, 2, 3]   # "3*x^3 + 2*x^2 + 3*x + 3"


def test_compute_without_gpu_fails_loudly(pkg):
    if pkg.binding.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(pkg.FheError) as ei:
        pkg.Plan(Q16, 4).forward(np.array([1, 2, 3, 4], dtype=np.uint64))
    assert ei.value.code == pkg.binding.FHE_E_NO_DEVICE
    assert "no CPU fallback" in str(ei.value)


def test_next_row_entry_points_validate_before_touching_the_gpu(pkg):
    """argument errors of the N1-N4 entry points are reported without a device"""
    L, B = pkg.load_library(), pkg.binding
    plan = pkg.Plan(Q16, 8)
    d = 16  # any non-NULL, 16-byte aligned fake device address: validation must fail first
    assert L.fhe_tn_mul_dev(3, d, d, d, 1, None) == B.FHE_E_BAD_N
    assert L.fhe_r_naive_mul_dev(1 << 20, d, d, d, 1, 0, 0, None) == B.FHE_E_BAD_N      # 2n would exceed 2^20
    assert L.fhe_bfv_tensor_dev(0, 16, 2, d, d, 1, None) == B.FHE_E_BAD_Q
    assert L.fhe_bfv_relinearize_dev(Q16, 16, Q16 - 1, d, d, d, 1, None) == B.FHE_E_BAD_Q  # pq < q
    assert L.fhe_tggsw_external_product_dev(64, 4, 65, d, d, d, 1, None) == B.FHE_E_INVALID
    assert L.fhe_rq_decompose_dev(Q16, 8, 1, 4, d, d, 1, None) == B.FHE_E_INVALID
    assert L.fhe_rq_div_round_dev(Q16, 0, d, d, 1, None) == B.FHE_E_INVALID
    assert L.fhe_rq_remodule_dev(0, d, d, 1, None) == B.FHE_E_BAD_Q
    assert L.fhe_glwe_key_switch_dev(plan.handle, 0, 2, 4, d, d, d, 1, 0, None) == B.FHE_E_INVALID
    assert L.fhe_glwe_key_switch_dev(plan.handle, 1, 2, 4, d, d, d, 1, B.FHE_OUT_EVALS, None) == B.FHE_E_INVALID  # key flag only
    assert L.fhe_glev_mul_dev(plan.handle, 1, 4, d, d, d, 1, 8, None) == B.FHE_E_INVALID                        # unknown flag bit
    assert L.fhe_tr_dot_dev(None, d, d, d, 2, 1, 0, None) == B.FHE_E_NULL
    assert L.fhe_rq_add_dev(plan.handle, None, d, d, 1, None) == B.FHE_E_NULL
    assert L.fhe_mul_div_round_dev(Q16, 8, d, 1, 0, d, 1, None) == B.FHE_E_BAD_Q           # den = 0
    # Zq::decompose is only defined where beta^l fits u32 and q / beta^l >= 1 (zq.rs:152,164-165), base 2
    # where l <= 64: outside, the reference panics (overflow / division by zero) — never a launch here
    assert L.fhe_rq_decompose_dev(Q16, 8, 2, 65, d, d, 1, None) == B.FHE_E_INVALID
    assert L.fhe_rq_decompose_dev(Q16, 8, 2, 64, None, d, 1, None) == B.FHE_E_NULL              # l = 64 itself is accepted
    assert L.fhe_rq_decompose_dev(Q61, 8, 16, 9, d, d, 1, None) == B.FHE_E_INVALID              # 16^9 overflows u32
    assert b"overflows u32" in L.fhe_last_error()
    assert L.fhe_rq_decompose_dev(Q16, 8, 4, 9, d, d, 1, None) == B.FHE_E_INVALID               # 4^9 > q: divisor 0
    assert b"divisor" in L.fhe_last_error()
    assert L.fhe_rq_decompose_dev(Q16, 8, 4, 8, None, d, 1, None) == B.FHE_E_NULL               # 4^8 = 65536 <= q: fine
    assert L.fhe_glwe_key_switch_dev(plan.handle, 1, 2, 65, d, d, d, 1, 0, None) == B.FHE_E_INVALID
    assert L.fhe_glwe_key_switch_dev(plan.handle, 1, 4, 9, d, d, d, 1, 0, None) == B.FHE_E_INVALID
    # resident key-switching key: sizes without a device, argument errors before any launch
    assert L.fhe_glwe_ksk_prepared_words(None, 1, 2, 4) == 0
    assert L.fhe_glwe_ksk_prepared_words(plan.handle, 0, 2, 4) == 0
    big = pkg.Plan(Q61, 4096)
    # arguments fhe_glwe_ksk_prepare_dev rejects have no prepared form: 0 words, and no error string is set
    L.fhe_ntt_plan_prepare(None)                                                          # leaves "plan is NULL" behind
    before = L.fhe_last_error()
    assert L.fhe_glwe_ksk_prepared_words(pkg.Plan(Q16, 8).handle, 1, 4, 9) == 0            # 4^9 > q: divisor 0 (zq.rs:164-165)
    assert L.fhe_glwe_ksk_prepared_words(big.handle, 1, 2, 65) == 0                        # beta = 2: l <= 64
    assert L.fhe_glwe_ksk_prepared_words(big.handle, 1, 70000, 2) == 0                     # beta^l overflows u32
    assert L.fhe_last_error() == before
    # the two-small-prime form exists unless FHE_EXT32=0 (read once per process): twice the key, or the key's size
    ext32 = os.environ.get("FHE_EXT32", "1")[:1] != "0"
    assert L.fhe_glwe_ksk_prepared_words(big.handle, 1, 2, 61) == (2 if ext32 else 1) * 61 * 2 * 4096
    assert L.fhe_glwe_ksk_prepared_words(big.handle, 2, 2, 61) == 2 * 61 * 3 * 4096       # k = 2: transforms modulo q
    assert L.fhe_glwe_ksk_prepared_words(big.handle, 1, 4, 8) == 8 * 2 * 4096
    assert L.fhe_glwe_ksk_prepare_dev(None, 1, 2, 4, d, d, None) == B.FHE_E_NULL
    # resident relinearisation key: three 27-bit primes where c2 * rlk over n terms stays below 2^82, 61-bit forms otherwise
    assert L.fhe_bfv_rlk_prepared_words(Q16, 8192, Q16 ** 3) == 6 * 8192                 # 17 + 51 + 13 bits
    assert L.fhe_bfv_rlk_prepared_words(Q16, 16384, Q16 ** 3) == 8 * 16384               # n > 8192: one 61-bit prime, key in halves
    assert L.fhe_bfv_rlk_prepared_words(786433, 4096, 786433 ** 3) == 8 * 4096           # 20 + 59 + 12 > 82 bits
    assert L.fhe_bfv_rlk_prepared_words(Q16, 512, Q16 ** 3) == 8 * 512                   # n < 1024: 61-bit path
    assert L.fhe_glwe_ksk_prepare_dev(plan.handle, 1, 2, 65, d, d, None) == B.FHE_E_INVALID
    assert L.fhe_glwe_key_switch_prepared_dev(plan.handle, 0, 2, 4, d, d, d, 1, None) == B.FHE_E_INVALID
    assert L.fhe_glwe_key_switch_prepared_dev(big.handle, 1, 2, 61, None, None, None, 0, None) == 0
    # empty batches are no-ops everywhere
    assert L.fhe_tn_mul_dev(8, None, None, None, 0, None) == 0
    assert L.fhe_bfv_mul_dev(Q16, 16, 2, Q16 * Q16 * Q16, None, None, None, 0, None) == 0
    assert L.fhe_tr_dot_dev(plan.handle, None, None, None, 2, 0, 0, None) == 0


def test_header_is_plain_c_and_the_c_example_links(pkg, tmp_path):
    """include/fhe_ntt.h must be consumable from C99 (the FFI of any language binds C), and
    examples/rq_mul.c must link against the library using nothing else"""
    import subprocess
    hdr = os.path.join(ROOT, "include", "fhe_ntt.h")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c", hdr])
    exe = str(tmp_path / "rq_mul")
    libdir = os.path.dirname(pkg.binding.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-O1", "-o", exe,
                           os.path.join(ROOT, "examples", "rq_mul.c"), "-L" + libdir, "-lfhe_ntt",
                           "-Wl,-rpath," + libdir])
    assert os.path.exists(exe)


def test_shard_range_of_the_c_abi_is_the_partition_sharding_py_uses(pkg):
    """fhe_shard_range (host-only): the block partition of SURVEY.md §8e, identical to the Python twin
    the torch.distributed drivers use; argument errors are codes, not crashes."""
    import ctypes

    L, B = pkg.load_library(), pkg.binding
    for total in (0, 1, 7, 8, 630, 65536, 65537):
        for world in (1, 2, 3, 4, 8):
            rows = []
            for r in range(world):
                b0, b1 = B.shard_range(total, world, r)
                assert (b0, b1) == pkg.sharding.shard_range(total, world, r)
                rows += list(range(b0, b1))
            assert rows == list(range(total))
    assert [B.shard_range(630, 8, r)[1] - B.shard_range(630, 8, r)[0] for r in range(8)] == [79] * 7 + [77]
    b, e = ctypes.c_size_t(), ctypes.c_size_t()
    assert L.fhe_shard_range(10, 0, 0, ctypes.byref(b), ctypes.byref(e)) == B.FHE_E_INVALID
    assert L.fhe_shard_range(10, 2, 2, ctypes.byref(b), ctypes.byref(e)) == B.FHE_E_INVALID
    assert L.fhe_shard_range(10, 2, 1, None, ctypes.byref(e)) == B.FHE_E_NULL


def test_plan_prepare_and_check_switch_need_no_gpu_to_fail_cleanly(pkg):
    L, B = pkg.load_library(), pkg.binding
    assert L.fhe_ntt_plan_prepare(None) == B.FHE_E_NULL
    if B.device_count() == 0:
        assert L.fhe_ntt_plan_prepare(pkg.Plan(Q16, 8).handle) == B.FHE_E_NO_DEVICE
    assert L.fhe_ntt_set_check_canonical(1) == 0 and L.fhe_ntt_set_check_canonical(0) == 0


def test_plan_arithmetic_is_decided_on_the_host(pkg):
    """fhe_ntt_plan_arithmetic (round 3): which exact form of Zq::mul the kernels run — no device needed.  The
    pseudo-Mersenne rule is the one fhe-study_amd/arith.py restates (pm_params); FHE_PM / FHE_EXT32 are read once per
    process, so the expectations follow the environment."""
    from fhe_study_amd.arith import pm_params

    L, B = pkg.load_library(), pkg.binding
    assert L.fhe_ntt_plan_arithmetic(None) == B.FHE_E_NULL
    pm_on = os.environ.get("FHE_PM", "1")[:1] != "0"
    ext_on = os.environ.get("FHE_EXT32", "1")[:1] != "0"
    mg_on = os.environ.get("FHE_MG", "1")[:1] != "0"
    cases = [
        (Q61, 65536, 2 if pm_on else 1),                      # 2^61 - 2^21 + 1: the headline modulus
        (2305843009210023937, 4096, 2 if pm_on else 1),       # 2^61 - 28 * 2^17 + 1
        (1152921504606584833, 4096, 2 if pm_on else 1),       # 2^60 - 2^18 + 1
        (2305843009208713217, 4096, 1),                       # 2^61 - 38 * 2^17 + 1: delta too large
        (0x1fffffffff000001, 4096, 1),                        # 2^61 - 2^24 + 1 (a CRT prime of zring.hip): not of the form
        (4611686018425815041, 4096, 0),                       # just below 2^62
        (Q16, 4096, 3 if ext_on else 1),
        (Q16, 64, 1),                                         # n below the 32-bit kernels' range
        (1073479681, 4096, 3 if ext_on else 1),               # a 30-bit prime: Harvey form of the 32-bit kernels
        (2147352577, 4096, 1),                                # 31 bits: past them
        (0x1ffffff900000001, 65536, 5 if mg_on else 1),       # q = 1 (mod 2^32): word Montgomery in the forward transforms (round 4)
        (0x3fff300000001, 4096, 5 if mg_on else 1),           # ... of 50 bits
        (0xff00000001, 16, 5 if mg_on else 1),                # ... of 40 bits, the smallest n the block kernels run
        (0xff00000001, 8, 1),                                 # ... below it: one thread per polynomial, Shoup
    ]
    for q, n, want in cases:
        assert (q - 1) % (2 * n) == 0
        plan = pkg.Plan(q, n)
        assert plan.arithmetic() == want, (q, n)
        assert (pm_params(q) is not None) == (want == 2 or (not pm_on and q in (Q61, 2305843009210023937, 1152921504606584833)))
    assert set(pkg.Plan.ARITH_NAMES) == {0, 1, 2, 3, 4, 5}  # 4: 2^62 <= q < 2^63, tests/test_round3_gpu.py; 5: tests/test_round4_gpu.py
