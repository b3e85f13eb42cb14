"""GPU parity of the preconditioner family of src/timestepper.F90:1745-1757 beyond the fused
block-Jacobi brick kernels: PCASM (restricted, overlap 1 and 2), subdomains of any size including
the reference's layout of one block per rank (sub_ptr = NULL), PCNONE -- each against the CPU
oracle's restatement through the C ABI, on preconditioner applications and on whole Krylov solves."""
import numpy as np
import pytest

from oracle import binding as ol
from waiwera_amd.cases import make_case, scaled

pytestmark = pytest.mark.gpu

KIND = {"w": 0, "we": 1, "wce": 2, "wsce": 5}


def relmax(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(b).max(), 1e-300)


def system(oracle, eos, dims, brick, one_block=False, lens=True, dt=5.0e4, **kw):
    from waiwera_amd.flow_simulation import FlowSimulation
    g, lm, prim, region = make_case(dims=dims, brick=brick, eos=eos, lens=lens, **kw)
    if one_block:
        lm.sub_ptr = None
    sim = FlowSimulation(lm, eos=eos)
    if one_block:
        lm.sub_ptr = np.array([0, lm.n_owned], dtype=np.int32)
    osim = ol.OracleSim(oracle, lm, KIND[eos])
    sim.set_regions(region); osim.set_regions(region)
    y = scaled(prim, region, eos).ravel().copy()
    yo = osim.yvec(y)
    assert osim.pre_eval(yo) == 0
    L = osim.lhs()
    err, f = osim.residual(yo, dt, L)
    err, J = osim.jacobian(yo, dt, L, f, mode=0)
    assert err == 0
    sim.set_jacobian_values(J)
    return lm, sim, osim, J, f


@pytest.mark.parametrize("eos,overlap,brick", [("we", 1, (4, 4, 2)), ("we", 2, (4, 4, 2)), ("w", 1, (4, 4, 4)),
                                               ("wce", 1, (4, 4, 2)), ("wsce", 1, (4, 2, 2))])
def test_asm_application_and_solve(oracle, eos, overlap, brick):
    lm, sim, osim, J, f = system(oracle, eos, (8, 8, 6), brick, lens=(eos == "we"))
    n = sim.num_dof
    sim.set_opts(pc_type="asm", asm_overlap=overlap)
    osim.set_asm(overlap)
    assert sim.pc_setup() == 0 and osim.pc_setup(J) == 0
    r = np.random.default_rng(11).normal(size=n)
    z = np.zeros(n)
    sim.pc_apply(r, z)
    assert relmax(z, osim.pc_apply(r)) < 1e-10
    # overlap changes the operator: not the block-Jacobi result
    sim.set_opts(pc_type="bjacobi")
    zb = np.zeros(n)
    sim.pc_apply(r, zb)
    assert relmax(z, zb) > 1e-6
    sim.set_opts(pc_type="asm", asm_overlap=overlap)
    for ksp, kt in (("bcgs", 0), ("gmres", 1)):
        sim.set_opts(ksp_type=ksp, ksp_rtol=1e-12)
        x = np.zeros(n)
        its, reason, rn = sim.ksp_solve(f, x)
        oreason, xo, oits, hist = osim.ksp_solve(J, f, ksp_type=kt, rtol=1e-12)
        assert reason > 0 and oreason > 0
        assert relmax(x, xo) < 1e-8
        assert abs(its - oits) <= max(2, oits // 10)
    sim.destroy(); osim.close()


@pytest.mark.parametrize("eos,one_block,pc", [("we", True, "bjacobi"), ("we", False, "bjacobi"), ("wce", True, "bjacobi"),
                                              ("we", False, "asm")])
def test_subdomains_larger_than_a_workgroup(oracle, eos, one_block, pc):
    """one block per rank (the reference's layout; sub_ptr = NULL) and 12 x 12 x 10 bricks of 1440 rows:
    the launch-per-level path, block Jacobi and ASM over it"""
    lm, sim, osim, J, f = system(oracle, eos, (24, 12, 10), (24, 12, 10) if one_block else (12, 12, 10), one_block=one_block)
    n = sim.num_dof
    if pc == "asm":
        sim.set_opts(pc_type="asm"); osim.set_asm(1)
    assert sim.pc_setup() == 0 and osim.pc_setup(J) == 0
    r = np.random.default_rng(12).normal(size=n)
    z = np.zeros(n)
    sim.pc_apply(r, z)
    assert relmax(z, osim.pc_apply(r)) < 1e-10
    sim.set_opts(ksp_rtol=1e-12)
    x = np.zeros(n)
    its, reason, rn = sim.ksp_solve(f, x)
    oreason, xo, oits, hist = osim.ksp_solve(J, f, rtol=1e-12)
    assert reason > 0 and oreason > 0
    assert relmax(x, xo) < 1e-8 and abs(its - oits) <= max(2, oits // 10)
    sim.destroy(); osim.close()


def test_no_preconditioner(oracle):
    lm, sim, osim, J, f = system(oracle, "w", (6, 6, 4), (3, 3, 2), lens=False, dt=10.0)   # diagonally dominant
    n = sim.num_dof
    sim.set_opts(pc_type="none", ksp_rtol=1e-10, ksp_max_its=2000)
    oracle.wo_sim_set_pc_none(osim.h, 1)
    r = np.random.default_rng(13).normal(size=n)
    z = np.zeros(n)
    sim.pc_apply(r, z)
    assert np.array_equal(z, r)
    x = np.zeros(n)
    its, reason, rn = sim.ksp_solve(f, x)
    oreason, xo, oits, hist = osim.ksp_solve(J, f, rtol=1e-10, maxits=2000)
    assert reason > 0 and oreason > 0 and relmax(x, xo) < 1e-6
    sim.destroy(); osim.close()


def test_time_step_with_asm_matches_the_oracle(oracle):
    """a whole backward-Euler step with the reference's default preconditioner on both sides"""
    from waiwera_amd.flow_simulation import FlowSimulation
    g, lm, prim, region = make_case(dims=(8, 8, 6), brick=(4, 4, 2), eos="we", lens=True)
    sim = FlowSimulation(lm, eos="we")
    osim = ol.OracleSim(oracle, lm, 1)
    sim.set_regions(region); osim.set_regions(region)
    sim.set_opts(pc_type="asm", ksp_rtol=1e-10, ftol_rel=1e-9)
    osim.set_asm(1)
    y = scaled(prim, region).ravel().copy()
    yo = osim.yvec(y)
    o = osim.opts(); o.ksp_rtol = 1e-10; o.ftol_rel = 1e-9
    r_o, k_o = osim.timestep(yo, 2.0e4, o)
    reason, nits, kits = sim.timestep(2.0e4, 2.0e4, y)
    assert reason > 0 and r_o > 0 and nits == r_o
    assert relmax(y, yo[: y.size]) < 1e-7
    assert np.array_equal(sim.regions(), osim.regions())
    sim.destroy(); osim.close()


@pytest.mark.parametrize("eos,brick", [("we", (4, 4, 2)), ("wce", (4, 4, 2))])
def test_bicgstab_with_merged_reductions(oracle, eos, brick, monkeypatch):
    """the form ranks > 1 run -- (S,T), (T,T), (S,S), (S,RP), (T,RP) in one reduction, (R,R) and (R,RP)
    derived from them -- forced onto one rank: same solution as the oracle's KSPBCGS, iteration
    count within rounding"""
    monkeypatch.setenv("WAI_BCGS_MERGED", "1")
    lm, sim, osim, J, f = system(oracle, eos, (8, 8, 6), brick, lens=(eos == "we"))
    n = sim.num_dof
    sim.set_opts(ksp_rtol=1e-12)
    x = np.zeros(n)
    its, reason, rn = sim.ksp_solve(f, x)
    oreason, xo, oits, hist = osim.ksp_solve(J, f, rtol=1e-12)
    assert reason > 0 and oreason > 0
    assert relmax(x, xo) < 1e-8 and abs(its - oits) <= max(2, oits // 10)
    sim.destroy(); osim.close()


@pytest.mark.parametrize("eos,brick,minc", [("we", (4, 4, 2), False), ("wce", (4, 4, 2), False), ("wce", (4, 4, 1), True),
                                            ("w", (4, 4, 4), False), ("wsce", (4, 2, 2), False)])
def test_bicgstab_fused_iteration(oracle, eos, brick, minc, monkeypatch):
    """The default iteration (WAI_BCGS=fused): merged reductions, the X / R / next-P updates in one pass that re-forms
    S = R - alpha V -- four launches; with WAI_BCGS_COMPOSE=1 the second fused launch forms S itself -- three.  Every
    expression is the one the five-launch merged form evaluates (WAI_BCGS=merged), so all of them must agree BIT FOR BIT:
    same iteration count, same residual norm, same solution; with the reductions finished by separate k_finalize launches
    (WAI_FIN_SEPARATE) too.  Against the oracle's KSPBCGS: within rounding.  eos w runs the generic k_pc (no composed
    operand), we k_pc_park, wce k_pc_wave, the MINC bricks k_pc_wave with short rows, wsce k_pc_rows<4>."""
    lm, sim, osim, J, f = system(oracle, eos, (8, 8, 6), brick, lens=(eos == "we"), **({"minc": True} if minc else {}))
    n = sim.num_dof
    sim.set_opts(ksp_rtol=1e-12)
    out = {}
    for tag, env in (("fused", {"WAI_BCGS": "fused", "WAI_BCGS_COMPOSE": "0"}), ("default", {}), ("merged", {"WAI_BCGS": "merged"}),
                     ("composed", {"WAI_BCGS": "fused", "WAI_BCGS_COMPOSE": "1"}),
                     ("fin_separate", {"WAI_BCGS": "fused", "WAI_FIN_SEPARATE": "1"}), ("petsc", {"WAI_BCGS": "petsc"})):
        for k in ("WAI_BCGS", "WAI_BCGS_COMPOSE", "WAI_FIN_SEPARATE", "WAI_BCGS_MERGED"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        x = np.zeros(n)
        k0 = sim.launch_stats()[0]
        its, reason, rn = sim.ksp_solve(f, x)
        out[tag] = (its, reason, rn, x, (sim.launch_stats()[0] - k0) / max(its, 1))
        assert reason > 0, (tag, its, reason, rn)
    oreason, xo, oits, hist = osim.ksp_solve(J, f, rtol=1e-12)
    assert oreason > 0
    its, _, rn, x, per = out["fused"]
    for tag in ("merged", "composed", "fin_separate", "default"):
        assert out[tag][0] == its and out[tag][2] == rn and np.array_equal(out[tag][3], x), (tag, out[tag][:3], its, rn)
    # launches per iteration (the speculative half of an iteration that is then not needed and the set-up add a few)
    kernel = sim.pc_kernel_name()
    can_compose = not kernel.startswith("k_pc<")
    assert 4 <= per <= 4 + 8.0 / its, (kernel, per)
    assert (3 if can_compose else 4) <= out["composed"][4] <= (3 if can_compose else 4) + 8.0 / its, (kernel, out["composed"][4])
    # the default: composed for k_pc_park on its 16-bit column indices and for k_pc_wave (round 5: measured faster), stored S elsewhere
    dflt = 3 if ("col16" in kernel or kernel.startswith("k_pc_wave")) else 4
    assert dflt <= out["default"][4] <= dflt + 8.0 / its, (kernel, out["default"][4])
    assert out["merged"][4] >= 5 and out["fin_separate"][4] >= 5 and 5 <= out["petsc"][4] <= 5 + 8.0 / its
    for tag in ("fused", "petsc"):
        assert relmax(out[tag][3], xo) < 1e-8 and abs(out[tag][0] - oits) <= max(2, oits // 10), (tag, out[tag][0], oits)
    sim.destroy(); osim.close()


@pytest.mark.parametrize("eos", ["we", "wce", "w"])
def test_ilu0_with_off_diagonal_fill(oracle, eos):
    """A cell graph WITH triangles: diagonal connections (i, j, k) - (i + 1, j + 1, k) added to a structured mesh make
    (i, j), (i + 1, j), (i + 1, j + 1) pairwise adjacent, so the IKJ elimination updates off-diagonal blocks inside a
    subdomain, ILU(0) is no longer DILU and the library takes the stored-factor kernels (k_ilu_factor, k_pc with the
    factor read back: the path of meshes with triangles in their cell graph, which the hexahedral / MINC tests never
    reach).  Rows have up to 9 blocks > the register-resident 8?  No: 6 + 2 diagonals + itself = 9 only in the interior of
    3-D meshes, so the mesh here is one layer thick (4 + 2 + 1 = 7).  Application and Krylov solve against the oracle."""
    import waiwera_amd.cases as cases
    from waiwera_amd.flow_simulation import FlowSimulation
    dims, brick = (10, 9, 1), (5, 3, 1)
    g, lm, prim, region = cases.make_case(dims=dims, brick=brick, eos=eos, lens=False, top_bc=False)
    ijk = np.asarray(lm.owned_ijk)
    idx = {(int(a), int(b)): q for q, (a, b, c) in enumerate(ijk)}
    fg_x = None
    fc = np.asarray(lm.face_cells)
    for f in range(lm.n_faces):     # a template: any x face between two owned cells
        a, b = fc[f]
        if a < lm.n_owned and b < lm.n_owned and abs(ijk[a][0] - ijk[b][0]) == 1:
            fg_x = np.asarray(lm.face_geom)[f].copy()
            break
    extra_c, extra_g = [], []
    for (i, j), q in idx.items():
        if (i + 1, j + 1) in idx:
            row = fg_x.copy()
            row[0] *= 0.3                                   # area
            row[1] = row[2] = 0.5 * np.hypot(10.0, 10.0)    # distances to the face
            row[3] = row[1] + row[2]
            extra_c.append((q, idx[(i + 1, j + 1)]))
            extra_g.append(row)
    lm.face_cells = np.concatenate([fc, np.array(extra_c, dtype=np.int32)]).astype(np.int32)
    lm.face_geom = np.concatenate([np.asarray(lm.face_geom), np.array(extra_g)])
    lm.n_faces = lm.face_cells.shape[0]
    sim = FlowSimulation(lm, eos=eos)
    osim = ol.OracleSim(oracle, lm, KIND[eos])
    sim.set_regions(region); osim.set_regions(region)
    assert sim.pc_kernel_name().startswith("k_pc<") and ",ilu," in sim.pc_kernel_name(), sim.pc_kernel_name()
    y = scaled(prim, region, eos).ravel().copy()
    yo = osim.yvec(y)
    dt = 5.0e4
    assert osim.pre_eval(yo) == 0
    L = osim.lhs()
    err, f = osim.residual(yo, dt, L)
    err, J = osim.jacobian(yo, dt, L, f, mode=0)
    assert err == 0
    sim.set_jacobian_values(J)
    n = sim.num_dof
    assert sim.pc_setup() == 0 and osim.pc_setup(J) == 0
    r = np.random.default_rng(3).normal(size=n)
    z = np.zeros(n)
    sim.pc_apply(r, z)
    assert relmax(z, osim.pc_apply(r)) < 1e-10
    sim.set_opts(ksp_rtol=1e-12)
    x = np.zeros(n)
    its, reason, rn = sim.ksp_solve(f, x)
    oreason, xo, oits, hist = osim.ksp_solve(J, f, rtol=1e-12)
    assert reason > 0 and oreason > 0
    assert relmax(x, xo) < 1e-8 and abs(its - oits) <= max(2, oits // 10), (its, oits)
    # the device's own FD Jacobian on this mesh equals the oracle's too (9-point rows through the assembly sweeps)
    assert sim.pre_eval(0.0, y) == 0
    fd = np.zeros(n)
    assert sim.residual(0.0, dt, y, L, fd) == 0 and relmax(fd, f) < 1e-11
    assert sim.jacobian(0.0, dt, y, L) == 0
    Jg = sim.jacobian_values()
    assert np.abs(Jg - J).max() <= 2e-5 * np.abs(J).max()
    sim.destroy(); osim.close()


@pytest.mark.parametrize("eos,pc", [("we", "bjacobi"), ("wce", "bjacobi"), ("we", "asm")])
def test_bcgsl(oracle, eos, pc):
    """BiCGStab(2) ("linear.type": "bcgsl") against the oracle's restatement"""
    lm, sim, osim, J, f = system(oracle, eos, (8, 8, 6), (4, 4, 2), lens=(eos == "we"))
    n = sim.num_dof
    sim.set_opts(ksp_type="bcgsl", pc_type=pc, ksp_rtol=1e-12)
    if pc == "asm":
        osim.set_asm(1)
    x = np.zeros(n)
    its, reason, rn = sim.ksp_solve(f, x)
    oreason, xo, oits, hist = osim.ksp_solve(J, f, ksp_type=2, rtol=1e-12)
    assert reason > 0 and oreason > 0 and its % 2 == 0
    assert relmax(x, xo) < 1e-8 and abs(its - oits) <= max(2, oits // 10)
    sim.destroy(); osim.close()


def test_lu_blocks(oracle):
    """"lu": exact block solves.  One block (sub_ptr = NULL): the preconditioner is the inverse, BiCGStab
    stops after one iteration at the dense solution; bricks: each block solved exactly"""
    import scipy.sparse as sp
    for one_block in (True, False):
        lm, sim, osim, J, f = system(oracle, "we", (6, 6, 4), (6, 6, 4) if one_block else (3, 3, 2), one_block=one_block)
        n = sim.num_dof
        rp, ci = osim.pattern()
        A = sp.bsr_matrix((J.reshape(-1, 2, 2), ci, rp), shape=(n, n)).toarray()
        sim.set_opts(pc_type="lu", ksp_rtol=1e-10)
        assert sim.pc_setup() == 0
        r = np.random.default_rng(14).normal(size=n)
        z = np.zeros(n)
        sim.pc_apply(r, z)
        ref = np.zeros(n)
        sub = [0, lm.n_owned] if one_block else list(lm.sub_ptr)
        for a, b in zip(sub[:-1], sub[1:]):
            ref[2 * a:2 * b] = np.linalg.solve(A[2 * a:2 * b, 2 * a:2 * b], r[2 * a:2 * b])
        assert relmax(z, ref) < 1e-6        # cond(A) ~ 6e10: an explicit inverse is good to cond x eps
        x = np.zeros(n)
        its, reason, rn = sim.ksp_solve(f, x)
        assert reason > 0 and relmax(x, np.linalg.solve(A, f)) < 1e-5
        if one_block:
            assert its <= 2
        sim.destroy(); osim.close()


def test_lgmres(oracle):
    """LGMRES ("linear.type": "lgmres"; restart 10 = 8 Krylov directions + 2 error approximations)
    against the oracle's restatement"""
    lm, sim, osim, J, f = system(oracle, "we", (8, 8, 6), (4, 4, 2))
    n = sim.num_dof
    sim.set_opts(ksp_type="lgmres", gmres_restart=10, ksp_rtol=1e-11, ksp_max_its=3000)
    x = np.zeros(n)
    its, reason, rn = sim.ksp_solve(f, x)
    oreason, xo, oits, hist = osim.ksp_solve(J, f, ksp_type=3, restart=10, rtol=1e-11, maxits=3000)
    assert reason > 0 and oreason > 0
    assert relmax(x, xo) < 1e-7 and abs(its - oits) <= max(3, oits // 8)
    sim.destroy(); osim.close()


@pytest.mark.timeout(120)
def test_lgmres_with_a_tiny_restart_and_the_restart_cap(oracle):
    """restart <= 2 leaves LGMRES no Krylov direction beside its two error approximations: it runs with
    one (restart 3) instead of spinning; a restart above the basis cap is refused, not silently cut"""
    lm, sim, osim, J, f = system(oracle, "we", (8, 8, 6), (4, 4, 2))
    n = sim.num_dof
    for restart in (1, 2, 3):
        sim.set_opts(ksp_type="lgmres", gmres_restart=restart, ksp_rtol=1e-6, ksp_max_its=400)
        x = np.zeros(n)
        its, reason, rn = sim.ksp_solve(f, x)
        assert reason != 0 and 0 < its <= 400, (restart, its, reason)
    from waiwera_amd.lib import WaiError
    with pytest.raises(WaiError, match="restart"):
        sim.set_opts(ksp_type="gmres", gmres_restart=41)
    sim.destroy(); osim.close()


@pytest.mark.parametrize("eos,pc,brick", [("we", "bjacobi", (4, 4, 3)), ("we", "asm", (4, 4, 2)), ("wce", "bjacobi", (4, 4, 3)),
                                          ("w", "bjacobi", (4, 4, 3))])
def test_iluk_sub_preconditioner(oracle, eos, pc, brick):
    """"sub_preconditioner": {"factor": {"levels": k}} (src/timestepper.F90:1716-1718, PCFactorSetLevels :1827),
    k = 1, 2, under block Jacobi and under PCASM: one application and whole Krylov solves against the oracle's
    ILU(k) (itself pinned on the fill-path definition, tests/test_oracle_linalg.py); ILU(k) is not ILU(0), and
    BiCGStab needs no more iterations with it"""
    lm, sim, osim, J, f = system(oracle, eos, (8, 8, 6), brick, lens=(eos == "we"))
    n = sim.num_dof
    r = np.random.default_rng(11).normal(size=n)
    ov = 1 if pc == "asm" else 0
    osim.set_asm(ov)
    sim.set_opts(pc_type=pc, asm_overlap=1, ilu_levels=0, ksp_rtol=1e-12)
    z0 = np.zeros(n)
    assert sim.pc_setup() == 0
    sim.pc_apply(r, z0)
    x = np.zeros(n)
    its0, reason, rn = sim.ksp_solve(f, x)
    assert reason > 0
    prev = its0
    for k in (1, 2):
        sim.set_opts(ilu_levels=k)
        osim.set_ilu_levels(k)
        assert sim.pc_setup() == 0 and osim.pc_setup(J) == 0
        z = np.zeros(n)
        sim.pc_apply(r, z)
        assert relmax(z, osim.pc_apply(r)) < 1e-10, k
        assert relmax(z, z0) > 1e-6
        x = np.zeros(n)
        its, reason, rn = sim.ksp_solve(f, x)
        oreason, xo, oits, hist = osim.ksp_solve(J, f, ksp_type=0, rtol=1e-12)
        assert reason > 0 and oreason > 0
        assert relmax(x, xo) < 1e-8
        assert abs(its - oits) <= max(2, oits // 10), (k, its, oits)
        assert its <= prev + 1, (k, its, prev)
        prev = its
        print(eos, pc, "ILU(%d):" % k, sim.pc_kernel_name(), "BiCGStab", its, "(oracle %d, ILU(0) %d)" % (oits, its0))
    # back to ILU(0): the fused path again, same result as before
    sim.set_opts(ilu_levels=0)
    osim.set_ilu_levels(0)
    assert sim.pc_setup() == 0
    z = np.zeros(n)
    sim.pc_apply(r, z)
    assert relmax(z, z0) < 1e-14
    sim.destroy(); osim.close()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("dims,brick", [((12, 12, 8), (4, 4, 2)), ((24, 24, 16), (2, 2, 2))],
                         ids=["one-finaliser", "two-finalisers"])
def test_a_lost_partial_sum_ends_the_solve_not_the_device(oracle, dims, brick):
    """The in-launch finalisation reads arrival off the data; a partial sum that never arrives (fault injected: workgroup 0
    of the next fused launch loses its store) must end in the finaliser's bounded wait, breakdown code 4 posted to the
    host and KSP_DIVERGED_NANORINF -- not in a hung device or a sum of stale data -- and the next solve, which empties the
    reduction slots first, must be untouched by it.  With 1 152 bricks the partials are summed in two slices by two
    finaliser workgroups: the one that gives up (slice 0) is not the one that posts, and the code must still arrive."""
    from waiwera_amd.lib import LIB
    lm, sim, osim, J, f = system(oracle, "we", dims, brick)
    assert (len(lm.sub_ptr) - 1 > 1024) == (brick == (2, 2, 2))
    n = sim.num_dof
    sim.set_opts(pc_type="bjacobi", ksp_type="bcgs", ksp_rtol=1e-10)
    assert sim.pc_setup() == 0
    x = np.zeros(n)
    its0, reason0, rn0 = sim.ksp_solve(f, x)
    assert reason0 > 0
    x0 = x.copy()
    assert LIB.wai_test_drop_partials(sim.h, 1) == 0
    x[:] = 0.0
    its, reason, rn = sim.ksp_solve(f, x)
    assert reason == -9, (its, reason, rn)
    assert "partial sum never arrived" in LIB.wai_last_error(sim.h).decode()
    x[:] = 0.0
    its2, reason2, rn2 = sim.ksp_solve(f, x)
    assert reason2 > 0 and its2 == its0 and np.array_equal(x, x0), (its2, its0, reason2)
    sim.destroy(); osim.close()


@pytest.mark.parametrize("dims,brick", [((12, 12, 8), (4, 4, 2)), ((20, 18, 6), (16, 16, 2))])
def test_brick_local_column_indices_same_bits(oracle, dims, brick, monkeypatch):
    """k_pc_park's 16-bit (segment, offset) column indices (IluSchedule::col16: 2 instead of 4 bytes per block) name the
    same columns in the same order as the int32 planes (WAI_NO_COL16=1, the path unstructured inputs with more than
    eight segments per brick keep): preconditioner applications and whole Krylov solves agree bit for bit; ragged
    bricks at the box's upper ends included"""
    lm, sim, osim, J, f = system(oracle, "we", dims, brick)
    n = sim.num_dof
    sim.set_opts(pc_type="bjacobi", ksp_type="bcgs", ksp_rtol=1e-10)
    assert sim.pc_setup() == 0
    assert sim.pc_kernel_name() == "k_pc_park<spmv,col16>"
    r = np.random.default_rng(21).normal(size=n)
    out = {}
    for tag in ("col16", "int32"):
        if tag == "int32":
            monkeypatch.setenv("WAI_NO_COL16", "1")
            assert sim.pc_kernel_name() == "k_pc_park<spmv>"
        z, x = np.zeros(n), np.zeros(n)
        sim.pc_apply(r, z)
        its, reason, rn = sim.ksp_solve(f, x)
        assert reason > 0
        out[tag] = (z, x, its, rn)
    assert np.array_equal(out["col16"][0], out["int32"][0]) and np.array_equal(out["col16"][1], out["int32"][1])
    assert out["col16"][2:] == out["int32"][2:]
    oreason, xo, oits, hist = osim.ksp_solve(J, f, ksp_type=0, rtol=1e-10)
    assert relmax(out["col16"][1], xo) < 1e-7
    sim.destroy(); osim.close()


def test_column_indices_fall_back_to_int32_when_a_brick_reaches_too_far(oracle, monkeypatch):
    """a brick whose rows reach more than eight 8 192-column segments (unstructured inputs) keeps the int32 column planes
    for the whole matrix: here the limit is lowered (WAI_COL16_MAX_SEG=1: the own brick only -- on this small mesh every
    other column falls into one window) so that the builder bails out on a structured mesh; the library must then run k_pc_park on the int32
    planes with the stored-S iteration and give the same solution as the default build of the same system"""
    out = {}
    for tag, env in (("col16", None), ("int32", "1")):
        if env:
            monkeypatch.setenv("WAI_COL16_MAX_SEG", env)
        lm, sim, osim, J, f = system(oracle, "we", (12, 12, 8), (4, 4, 2))
        sim.set_opts(pc_type="bjacobi", ksp_type="bcgs", ksp_rtol=1e-10)
        assert sim.pc_setup() == 0
        out[tag] = [sim.pc_kernel_name()]
        x = np.zeros(sim.num_dof)
        k0 = sim.launch_stats()[0]
        its, reason, rn = sim.ksp_solve(f, x)
        out[tag] += [its, reason, rn, x.copy(), (sim.launch_stats()[0] - k0) / max(its, 1)]
        sim.destroy(); osim.close()
    assert out["col16"][0] == "k_pc_park<spmv,col16>" and out["int32"][0] == "k_pc_park<spmv>"
    assert out["col16"][2] > 0 and out["int32"][2] > 0
    assert out["col16"][1] == out["int32"][1] and out["col16"][3] == out["int32"][3] and np.array_equal(out["col16"][4], out["int32"][4])
    assert out["col16"][5] < 3.5 and out["int32"][5] > 3.9       # composed (three launches) only on the 16-bit indices


@pytest.mark.parametrize("compose", ["0", None])
def test_bicgstab_iteration_is_four_launches_and_no_copy(oracle, compose, monkeypatch):
    """(three launches with the default of the 2 x 2 kernel since round 5: S = R - alpha V formed inside the second fused
    launch; WAI_BCGS_COMPOSE=0 is the stored-S form described here)
    one rank, fused block-Jacobi path: fused A*P + ILU(0) solve (+ (V,RP), alpha), S update, fused A*S + solve (+ the
    five merged inner products, omega, (R,R), rho, beta, the posted norm), X / R / next-P update in one pass -- every
    reduction finished inside its producer, the residual norm posted to pinned host memory: wai_launch_stats counts 4
    kernels per iteration (+ the speculative half iteration that is thrown away and the solve's set-up) and no copy per
    iteration"""
    if compose is not None:
        monkeypatch.setenv("WAI_BCGS_COMPOSE", compose)
    else:
        monkeypatch.delenv("WAI_BCGS_COMPOSE", raising=False)
    per = 4 if compose == "0" else 3
    lm, sim, osim, J, f = system(oracle, "we", (12, 12, 8), (4, 4, 2))
    n = sim.num_dof
    sim.set_opts(pc_type="bjacobi", ksp_type="bcgs", ksp_rtol=1e-10)
    assert sim.pc_setup() == 0
    x = np.zeros(n)
    k0, c0 = sim.launch_stats()
    its, reason, rn = sim.ksp_solve(f, x)
    k1, c1 = sim.launch_stats()
    assert reason > 0 and its >= 20
    assert per * its <= k1 - k0 <= per * its + 8, (its, k1 - k0)
    assert c1 - c0 <= 7, (its, c1 - c0)        # set-up only: RP = R, P = R, the initial norm, the staged vectors
    oreason, xo, oits, hist = osim.ksp_solve(J, f, ksp_type=0, rtol=1e-10)
    assert relmax(x, xo) < 1e-7 and abs(its - oits) <= max(2, oits // 10)
    sim.destroy(); osim.close()
