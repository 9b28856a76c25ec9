"""oracle — CPU restatement of the reference NTT path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package; the product (fhe-study_amd/) never does.
"""
from .oracle import Oracle, build_oracle, load_oracle  # noqa: F401
