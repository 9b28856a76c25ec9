"""oracle/ntt_oracle_py.py — pure-Python twin of oracle/ntt_oracle.c.

TEST INFRASTRUCTURE ONLY (see oracle/ntt_oracle.c header): used to generate the
fixtures under tests/golden/ (tests/golden/make_golden.py) and to cross-check
the C restatement with arbitrary-precision integers.  Small sizes only — these
are Python loops.

Citations are reference paths (arnaucube/fhe-study), relative to its root.
"""


def zq_add(q, a, b):
    """arith/src/zq.rs:219-231"""
    v = a + b
    return v - q if v >= q else v


def zq_sub(q, a, b):
    """arith/src/zq.rs:259-276"""
    return a - b if a >= b else (q + a) - b


def zq_mul(q, a, b):
    """arith/src/zq.rs:315-328 (u128 %)"""
    return (a * b) % q


def exp_mod(q, x, k):
    """arith/src/ntt.rs:164-179"""
    r = 1
    x %= q
    while k > 0:
        if k % 2 == 1:
            r = (r * x) % q
        x = (x * x) % q
        k //= 2
    return r


def inv_mod(q, x):
    """arith/src/ntt.rs:182-185"""
    return exp_mod(q, x, q - 2)


def primitive_root_of_unity(q, n2):
    """arith/src/ntt.rs:115-131; n2 = order of the root (callers pass 2n).
    Raises where the reference panics."""
    if n2 <= 0 or n2 & (n2 - 1):
        raise ValueError("n must be a power of two")
    if (q - 1) % n2 != 0:
        raise ValueError("(q-1) % n != 0")
    k = 1
    while k < q:
        w = exp_mod(q, k, (q - 1) // n2)
        if exp_mod(q, w, n2 // 2) != 1:
            return w
        k += 1
    raise ValueError("No primitive root of unity")


def bitrev(i, log_n):
    r = 0
    for b in range(log_n):
        r |= ((i >> b) & 1) << (log_n - 1 - b)
    return r


def roots_of_unity(q, n, w):
    """arith/src/ntt.rs:133-147"""
    log_n = n.bit_length() - 1
    return [exp_mod(q, w, bitrev(i, log_n)) for i in range(n)]


def roots_of_unity_inv(q, n, r):
    """arith/src/ntt.rs:149-161"""
    return [inv_mod(q, x) for x in r]


def roots(q, n):
    """arith/src/ntt.rs:20-38 → (roots, roots_inv, n_inv, psi)"""
    if n < 2:
        raise ValueError("n < 2 is degenerate in the reference (ntt.rs:139)")
    n_inv = inv_mod(q, n)
    psi = primitive_root_of_unity(q, 2 * n)
    r = roots_of_unity(q, n, psi)
    ri = roots_of_unity_inv(q, n, r)
    return r, ri, n_inv, psi


def ntt(q, n, rts, a):
    """arith/src/ntt.rs:44-73"""
    r = list(a)
    t, m = n // 2, 1
    while m < n:
        k = 0
        for i in range(m):
            S = rts[m + i]
            for j in range(k, k + t):
                U = r[j]
                V = zq_mul(q, r[j + t], S)
                r[j] = zq_add(q, U, V)
                r[j + t] = zq_sub(q, U, V)
            k += 2 * t
        t //= 2
        m *= 2
    return r


def intt(q, n, rts_inv, n_inv, a):
    """arith/src/ntt.rs:78-110"""
    r = list(a)
    t, m = 1, n // 2
    while m > 0:
        k = 0
        for i in range(m):
            S = rts_inv[m + i]
            for j in range(k, k + t):
                U = r[j]
                V = r[j + t]
                r[j] = zq_add(q, U, V)
                r[j + t] = zq_mul(q, zq_sub(q, U, V), S)
            k += 2 * t
        t *= 2
        m //= 2
    return [zq_mul(q, x, n_inv) for x in r]


def rq_mul(q, n, a, b):
    """arith/src/ring_nq.rs:586-607 → (c, c_evals, a_evals, b_evals)"""
    rts, rts_inv, n_inv, _ = roots(q, n)
    A = ntt(q, n, rts, a)
    B = ntt(q, n, rts, b)
    C = [zq_mul(q, x, y) for x, y in zip(A, B)]
    c = intt(q, n, rts_inv, n_inv, C)
    return c, C, A, B


def naive_negacyclic_mul(q, n, a, b):
    """arith/src/ring_n.rs:265-292 (+ X^N+1 fold) then to_rq, ring_nq.rs:116-129"""
    res = [0] * (2 * n - 1)
    for i in range(n):
        for j in range(n):
            res[i + j] += a[i] * b[j]
    out = res[:n]
    for i in range(n, 2 * n - 1):
        out[i - n] -= res[i]
    return [x % q for x in out]


_M64 = (1 << 64) - 1


def splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & _M64
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & _M64
    return x ^ (x >> 31)


def fill_synthetic(q, seed, first_index, count):
    """SURVEY.md §8d generator: mulhi64(splitmix64(seed ^ idx), q)"""
    return [(splitmix64(seed ^ (first_index + i)) * q) >> 64 for i in range(count)]
