"""CPU oracle: a plain-C restatement of the reference's algorithm for the hot path, plus its ctypes binding.
TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as the
checker / reported baseline; nothing under waiwera_amd/ or include/ may touch it."""
