import os
import subprocess
import sys

import pytest

import glob
import tempfile

os.environ.setdefault("OMP_NUM_THREADS", "1")  # the oracle is the checker: keep it deterministic
# simulation output files (output.filename of the reference's input files) go to a scratch
# directory, never beside the golden inputs
os.environ.setdefault("WAIWERA_OUTPUT_DIR", tempfile.mkdtemp(prefix="waiwera_amd_out_"))

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure). Built on demand with gcc."""
    so = os.path.join(ROOT, "oracle", "liboracle.so")
    srcs = glob.glob(os.path.join(ROOT, "oracle", "*.c")) + glob.glob(os.path.join(ROOT, "oracle", "*.h"))
    if (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")],
                              stdout=subprocess.DEVNULL)
    from oracle import binding as oracle_lib
    return oracle_lib.load(so)
