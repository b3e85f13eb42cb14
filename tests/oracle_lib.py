"""ctypes binding of the CPU oracle (oracle/binding.py), under the name the tests have always used."""
from oracle.binding import *  # noqa: F401,F403
from oracle.binding import C, d, pd, pi, i32  # noqa: F401
