"""RCCL plumbing on ONE GPU: a 1-rank communicator whose only neighbour is itself.  The halo
cells of a 2-rank partition's rank-0 mesh are filled by ncclSend/ncclRecv to self, which
exercises pack -> group send/recv -> unpack and the all-reduce path of the library exactly as
an N-rank run does (multi-GPU boxes are not available to the tests)."""
import numpy as np
import pytest

from waiwera_amd import mesh as M

pytestmark = pytest.mark.gpu


def test_self_halo_exchange_and_allreduce():
    from waiwera_amd import lib as wl
    from waiwera_amd.flow_simulation import FlowSimulation
    g = M.StructuredGrid((8, 8, 8), part=(2, 1, 1), brick=(4, 4, 4))
    lm = g.local_mesh(0, top_bc=([1.0e5, 20.0], 1))
    assert lm.n_halo > 0 and list(lm.nbr_ranks) == [1]
    lm.nbr_ranks = np.array([0], dtype=np.int32)     # talk to myself
    sim = FlowSimulation(lm, eos="we")
    sim.comm_init(0, 1, wl.comm_unique_id())
    rng = np.random.default_rng(0)
    for dof in (1, 2):
        v = np.zeros(lm.n_prim * dof)
        v[: lm.n_owned * dof] = rng.normal(size=lm.n_owned * dof)
        rc = wl.LIB.wai_halo_exchange(sim.h, v.ctypes.data, dof)
        assert rc == 0, wl.LIB.wai_last_error(sim.h)
        want = v[: lm.n_owned * dof].reshape(-1, dof)[lm.send_idx].ravel()
        assert np.array_equal(v[lm.n_owned * dof:], want)
    sim.destroy()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("ranks,delay_us", [(2, 0), (4, 0), (8, 0), (8, 100)])
def test_async_transport_selftest(ranks, delay_us):
    """the stream-asynchronous stand-in for librccl that the multi-rank tests run on (tests/loopback_rccl/async_rccl.hip),
    by itself: N processes on this one GPU, grouped sends / receives of changing lengths (8 bytes to three mailbox
    chunks) to both ring neighbours and sum / max / min all-reduces, filled and checked by kernels on two
    event-ordered streams, no host synchronisation until the end; no wrong word may arrive"""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "loopback_rccl", "async_selftest")
    assert os.path.exists(exe), "build first: python __graft_entry__.py"
    env = dict(os.environ, WAI_ASYNC_RCCL_DELAY_US=str(delay_us), WAI_ASYNC_RCCL_TIMEOUT_S="30")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if ranks >= 7:
        env.setdefault("GPU_MAX_HW_QUEUES", "1")     # 8 x 4 hardware queues oversubscribe the device's queue slots (tests/test_hip_multirank.py::_own_cus)
    out = subprocess.run([exe, str(ranks), "300"], env=env, capture_output=True, text=True, timeout=500)
    assert out.returncode == 0 and "PASSED" in out.stdout, (out.stdout[-2000:], out.stderr[-2000:])


@pytest.mark.timeout(600)
def test_toolchain_on_this_box_builds_and_runs_a_kernel(tmp_path):
    """__graft_entry__.build() finds the shipped libraries current (source digest) and compiles nothing on a GPU box; this
    compiles on THE BOX -- its hipcc against its own runtime -- the asynchronous transport and its stress test from their
    sources and runs the result on two ranks, so a toolchain / runtime mismatch between the authoring container and the box
    would show here and not first in the field"""
    import os
    import shutil
    import subprocess
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this box")
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "loopback_rccl")
    for f in ("async_rccl.hip", "async_selftest.hip"):
        shutil.copy(os.path.join(src, f), tmp_path)
    flags = ["-O2", "-std=c++17", "--offload-arch=gfx950"]
    subprocess.run([hipcc] + flags + ["-fPIC", "-shared", "-o", "libasync_rccl.so", "async_rccl.hip", "-lrt"], cwd=tmp_path, check=True, timeout=300)
    subprocess.run([hipcc] + flags + ["-o", "async_selftest", "async_selftest.hip", "-L.", "-lasync_rccl", "-Wl,-rpath,$ORIGIN", "-lrt"],
                   cwd=tmp_path, check=True, timeout=300)
    env = dict(os.environ, WAI_ASYNC_RCCL_TIMEOUT_S="30")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    out = subprocess.run([os.path.join(tmp_path, "async_selftest"), "2", "100"], env=env, capture_output=True, text=True, timeout=200)
    assert out.returncode == 0 and "PASSED" in out.stdout, (out.stdout[-1000:], out.stderr[-1000:])
