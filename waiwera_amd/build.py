"""Builds libwaiwera_hip.so for gfx950 with hipcc (in-tree, next to the sources).

    python -m waiwera_amd.build            # build if stale
    python -m waiwera_amd.build --force
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libwaiwera_hip.so")
SOURCES = ["capi.hip", "krylov.hip", "pc_setup.hip", "network.hip", "measure.hip", "kernels_assembly.hip", "kernels_linalg.hip", "comm.cpp"]
import glob  # noqa: E402
# every header any source could include: a stale object after a header edit is worse than a rebuild
HEADERS = sorted(os.path.basename(h) for pat in ("*.h", "*.hpp") for h in glob.glob(os.path.join(CSRC, pat))) + \
    [os.path.join("..", "..", "include", "waiwera_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result",
         "-Wno-unused-value"] + os.environ.get("WAI_EXTRA_HIPCC_FLAGS", "").split()


# The assembly and EOS kernels are memory-bound and their finite differences subtract nearly equal sums:
# no FMA contraction there, so that a product is rounded the same way wherever the compiler meets it --
# the oracle is built with -ffp-contract=off too (oracle/Makefile), and k_jacobian / k_jacobian_park then
# give bit-identical blocks for every EOS (with contraction 0.5 % of the eos we entries differed by 1 ulp
# of the residual, 2.5e-8 of the entry: enough to send a failing time step of the bench window down another
# Newton path).  WAI_ASM_CONTRACT=1 builds with the compiler's default contraction.
PER_FILE = {} if os.environ.get("WAI_ASM_CONTRACT") == "1" else {"kernels_assembly.hip": ["-ffp-contract=off"]}


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs = []
    procs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(CSRC, os.path.splitext(s)[0] + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            cmd = [hipcc] + FLAGS + PER_FILE.get(s, []) + ["-x", "hip", "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError("hipcc failed on %s" % s)
        if verbose and out:
            sys.stderr.write(out.decode())
    if force or procs or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
