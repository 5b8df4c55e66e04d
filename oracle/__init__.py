"""CPU oracle for the AttnLRP hot path -- TEST INFRASTRUCTURE ONLY.

This package is a from-scratch CPU (PyTorch-on-CPU, fp32/fp64) restatement of the
reference's LRP arithmetic for the path named in BASELINE.json.  It is the *checker*:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it.  Nothing under ``lrp-explains-transformers_amd/`` imports it, and the product
path raises if the HIP library is missing instead of falling back to this code.

Parity status: PINNED.  Every function here was checked in the build container against
the imported reference (``PYTHONPATH=/root/reference``) by ``tests/golden/make_golden.py``
which also froze the reference's outputs as fixtures under ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` re-checks the oracle against those fixtures on every run.
"""
from . import rules, llama  # noqa: F401
