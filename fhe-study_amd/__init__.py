"""fhe-study_amd — MI355X-native negacyclic NTT engine behind arith::NTT / arith::Rq.

Layout:
  csrc/        HIP kernels (gfx950) + the C ABI of include/fhe_ntt.h → libfhe_ntt.so
  binding.py   ctypes plumbing over the C ABI
  arith.py     host mirror of the reference's RingParam / Rq / NTT surface
  bfv.py       RLWE::tensor / RLWE::mul (bfv/src/lib.rs) over the exact-product rows
  tfhe.py      Tn x Tn and TGGSW x TGLWE (ring_torus.rs, tfhe/src/tggsw.rs)
  host/        the same mirror in C++ (arith.hpp), for compiled callers

The directory name carries a hyphen (repo convention); import it as
`fhe_study_amd` (fhe_study_amd.py at the repo root aliases this package).
"""
from . import binding  # noqa: F401
from .binding import FheError, Plan, build, load_library  # noqa: F401
from .arith import NTT, RingParam, Rq, mul, mul_mut  # noqa: F401


def __getattr__(name):
    # `sharding` needs torch.distributed; keep it off the import path of torch-free users
    if name in ("sharding", "bfv", "tfhe"):
        import importlib

        return importlib.import_module(__name__ + "." + name)
    raise AttributeError(name)

Q61 = 2305843009211596801  # 2^61 - 2^21 + 1, the engine's headline modulus
Q16 = 65537                # the modulus of every reference test
