"""Shared seeded test problems: the synthetic workloads of SURVEY.md section 8d live in the package
(waiwera_amd.cases, also bench.py's); the tests use them at sizes the oracle finishes in seconds."""
from waiwera_amd.cases import make_case, scaled  # noqa: F401
