"""Host-side mirror of the reference's flow_simulation_type (src/flow_simulation.F90), i.e. the
concrete ode_type (src/ode.F90:39-108) whose hot loops now run as HIP kernels.

Method names, argument order and the `err` convention follow the reference's type-bound
procedures (lhs, rhs, pre_eval, pre_iteration, pre_timestep, pre_retry_timestep,
post_linesearch, setup_jacobian); the SNES/KSP slots the reference fills with PETSc objects
(src/timestepper.F90:1552-1836) are the residual / jacobian / ksp_solve / newton_step methods.
Vectors are numpy arrays (host) or torch tensors (device): both are passed by address.
"""
import ctypes as C

import numpy as np

from . import lib as _lib
from .lib import LIB, WaiError


class FlowSimulation:
    def __init__(self, mesh, eos="we", opts=None, device=0, temperature=20.0,
                 relperm=("linear", [0.0, 1.0, 0.0, 1.0]), capillary=("zero", []), thermo="iapws",
                 permeability_modifier=None):
        self.mesh = mesh
        self.eos_name = eos
        self._keep = dict(
            face_cells=_lib._i32(mesh.face_cells), face_geom=_lib._f64(mesh.face_geom),
            cell_geom=_lib._f64(mesh.cell_geom), rock=_lib._f64(mesh.rock),
            sub_ptr=_lib._i32(mesh.sub_ptr) if mesh.sub_ptr is not None else None)
        k = self._keep
        md = _lib.MeshDesc()
        md.n_owned, md.n_halo, md.n_bc, md.n_faces = mesh.n_owned, mesh.n_halo, mesh.n_bc, mesh.n_faces
        md.face_cells = k["face_cells"].ctypes.data_as(_lib.pi)
        md.face_geom = k["face_geom"].ctypes.data_as(_lib.pd)
        md.cell_geom = k["cell_geom"].ctypes.data_as(_lib.pd)
        md.rock = k["rock"].ctypes.data_as(_lib.pd)
        if k["sub_ptr"] is not None:
            md.n_sub = k["sub_ptr"].size - 1
            md.sub_ptr = k["sub_ptr"].ctypes.data_as(_lib.pi)
        self.eos_desc = _lib.eos_desc(eos, temperature, relperm, capillary, thermo=thermo,
                                      permeability_modifier=permeability_modifier)
        self.opts = opts or _lib.default_opts()
        h = C.c_void_p()
        rc = LIB.wai_ctx_create(C.byref(md), C.byref(self.eos_desc), C.byref(self.opts), device, C.byref(h))
        self.h = h
        if rc != 0:
            msg = LIB.wai_last_error(h).decode() if h else "?"
            raise WaiError("wai_ctx_create failed (%d): %s" % (rc, msg))
        # "table" curves (before the boundary fluid is evaluated in wai_set_bc)
        for spec, keys in ((relperm, ((0, "liquid", [[0, 0], [1, 1]]), (1, "vapour", [[0, 0], [1, 1]]))),
                           (capillary, ((2, "pressure", [[0, 0], [1, 0]]),))):
            if spec[0] == "table":
                for which, key, default in keys:
                    xy = _lib._f64(spec[1].get(key, default))
                    self._chk(LIB.wai_set_curve_table(h, which, _lib.INTERP[spec[1].get("interpolation", "linear")],
                                                      len(xy), xy.ctypes.data_as(_lib.pd)), "set_curve_table")
        self.num_primary_variables = LIB.wai_block_size(h)
        self.fluid_dof = LIB.wai_num_fluid_dof(h)
        self.n_owned, self.n_prim, self.n_local = mesh.n_owned, mesh.n_prim, mesh.n_local
        self.num_dof = self.n_owned * self.num_primary_variables
        self.num_tracers, self.auxiliary = 0, False
        self.time = 0.0
        if mesh.n_bc:
            bp, br = _lib._f64(mesh.bc_primary), _lib._i32(mesh.bc_region)
            self._chk(LIB.wai_set_bc(h, bp.ctypes.data_as(_lib.pd), br.ctypes.data_as(_lib.pi)), "set_bc")
        if mesh.n_src:
            sc, sr = _lib._i32(mesh.src_cell), _lib._f64(mesh.src_rate)
            se, sk = _lib._f64(mesh.src_enthalpy), _lib._i32(mesh.src_component)
            self._chk(LIB.wai_set_sources(h, sc.size, sc.ctypes.data_as(_lib.pi), sr.ctypes.data_as(_lib.pd),
                                          se.ctypes.data_as(_lib.pd), sk.ctypes.data_as(_lib.pi)), "set_sources")
        if mesh.nbr_ranks is not None and len(mesh.nbr_ranks):
            nr, sp = _lib._i32(mesh.nbr_ranks), _lib._i32(mesh.send_ptr)
            si, rp = _lib._i32(mesh.send_idx), _lib._i32(mesh.recv_ptr)
            self._chk(LIB.wai_set_halo(h, nr.size, nr.ctypes.data_as(_lib.pi), sp.ctypes.data_as(_lib.pi),
                                       si.ctypes.data_as(_lib.pi), rp.ctypes.data_as(_lib.pi)), "set_halo")

    def update_rock(self, field, cells, values):
        """rock controls: one field of the rock record (0..2 permeability, 3 wet, 4 dry conductivity, 5 porosity,
        6 density, 7 specific heat) on the listed local cells, before a try"""
        ci, v = _lib._i32(cells), _lib._f64(np.broadcast_to(np.asarray(values, dtype=np.float64), np.shape(cells)))
        self._chk(LIB.wai_update_rock(self.h, int(field), ci.size, ci.ctypes.data_as(_lib.pi), v.ctypes.data_as(_lib.pd)), "update_rock")
        self.mesh.rock[ci, field] = v

    def set_source_rates(self, rate=None, enthalpy=None):
        """new rates / enthalpies of the sources in force (what table controls do before a try)"""
        r = _lib._f64(rate) if rate is not None else None
        e = _lib._f64(enthalpy) if enthalpy is not None else None
        for a in (r, e):
            if a is not None and a.size != self.mesh.n_src:
                raise ValueError("one value per source")
        self._chk(LIB.wai_update_sources(self.h, r.ctypes.data_as(_lib.pd) if r is not None else None,
                                         e.ctypes.data_as(_lib.pd) if e is not None else None), "update_sources")

    def set_source_controls(self, records):
        """state-dependent controls, one dict per source (waiwera_amd.lib.source_controls) or None"""
        if records is None:
            self._chk(LIB.wai_set_source_controls(self.h, None), "set_source_controls")
            return
        if len(records) != self.mesh.n_src:
            raise ValueError("one control record per source")
        self._chk(LIB.wai_set_source_controls(self.h, _lib.source_controls(records)), "set_source_controls")

    def set_source_network(self, spec):
        """groups and reinjectors (include/waiwera_hip.h, wai_set_source_network).  spec: dict(rate_specified,
        enthalpy_specified per source; groups [dict(inputs=[(kind, index)], scaling 0 | 1, limits=[(type, limit)])];
        reinjectors [dict(input=(kind, index), outputs=[dict(flow 1 | 2, out=(kind, index), rate, proportion,
        enthalpy)], overflow=(kind, index))]); kinds 0 none, 1 source, 2 group, 3 reinjector"""
        g, r = spec["groups"], spec["reinjectors"]
        self._net_keep, args = _lib.network_arrays(spec)
        self._chk(LIB.wai_set_source_network(self.h, *args), "set_source_network")
        self._net_sizes = (len(g), len(r))

    def set_source_global_index(self, n_global, global_index):
        """a source network across ranks: the global index of each of this rank's sources (the description given
        to set_source_network is then numbered globally, the same on every rank)"""
        gi = _lib._i32(global_index)
        self._gidx_keep = gi
        self._chk(LIB.wai_set_source_global_index(self.h, int(n_global), gi.ctypes.data_as(_lib.pi)), "set_source_global_index")

    def source_network(self):
        """(groups (n, 6): rate, enthalpy, water_rate, water_enthalpy, steam_rate, steam_enthalpy;
        reinjectors (n, 8): output water / steam rate, overflow rate, enthalpy, water rate, water enthalpy, steam
        rate, steam enthalpy) after the last residual evaluation"""
        ng, nr = getattr(self, "_net_sizes", (0, 0))
        G, R = np.zeros((max(ng, 1), 6)), np.zeros((max(nr, 1), 8))
        self._chk(LIB.wai_get_source_network(self.h, G.ctypes.data_as(_lib.pd), R.ctypes.data_as(_lib.pd)), "get_source_network")
        return G[:ng], R[:nr]

    def set_network_couplings(self, on=True, in_preconditioner=True):
        """the network's Jacobian blocks (flow_simulation_modify_jacobian, flow_simulation.F90:3023-3084) on / off;
        in_preconditioner: also inside the factor's pattern, as PETSc factors the widened matrix (default), or in the
        operator only"""
        self._chk(LIB.wai_set_network_couplings(self.h, (2 if in_preconditioner else 1) if on else 0), "set_network_couplings")

    def network_couplings(self):
        """(cells (m,), E (ml, m, bs, bs)): d R(cell i) / d y(cell j) through the network pass, of the last
        wai_jacobian (wai_get_network_couplings); m = 0 without a network or when E vanishes.  One rank: ml = m.
        A network on several ranks: the columns are the network's cells of all ranks ordered by (owner, cell), the
        rows this rank's own (cells >= 0, in that order); another rank's cell reads -1 - owner"""
        m = C.c_int(0)
        self._chk(LIB.wai_get_network_couplings(self.h, C.byref(m), None, None), "get_network_couplings")
        bs = self.num_primary_variables
        cells = np.zeros(m.value, dtype=np.int32)
        if not m.value:
            return cells, np.zeros((0, 0, bs, bs))
        self._chk(LIB.wai_get_network_couplings(self.h, C.byref(m), cells.ctypes.data_as(_lib.pi), None), "get_network_couplings")
        E = np.zeros((int((cells >= 0).sum()), m.value, bs, bs))
        if E.size:
            self._chk(LIB.wai_get_network_couplings(self.h, C.byref(m), None, E.ctypes.data_as(_lib.pd)), "get_network_couplings")
        return cells, E

    def separator_enthalpies(self, pressure):
        hf, hg = C.c_double(0.0), C.c_double(0.0)
        self._chk(LIB.wai_separator_enthalpies(self.h, pressure, C.byref(hf), C.byref(hg)), "separator_enthalpies")
        return hf.value, hg.value

    def source_rates(self):
        """(rate, enthalpy) of every source on the current fluid"""
        r, e = np.zeros(self.mesh.n_src), np.zeros(self.mesh.n_src)
        if self.mesh.n_src:
            self._chk(LIB.wai_get_source_rates(self.h, r.ctypes.data_as(_lib.pd), e.ctypes.data_as(_lib.pd)), "get_source_rates")
        return r, e

    # ------------------------------------------------------------------------------------------
    def _chk(self, rc, what):
        if rc < 0:
            raise WaiError("%s failed (%d): %s" % (what, rc, LIB.wai_last_error(self.h).decode()))
        return rc

    def destroy(self):
        if self.h:
            LIB.wai_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass

    def set_opts(self, **kw):
        for k, v in kw.items():
            if k == "ksp_type" and isinstance(v, str):
                v = _lib.KSP[v]
            if k == "pc_type" and isinstance(v, str):
                v = _lib.PC[v]
            setattr(self.opts, k, v)
        self._chk(LIB.wai_set_opts(self.h, C.byref(self.opts)), "set_opts")

    def comm_init(self, rank, nranks, unique_id):
        self._chk(LIB.wai_comm_init(self.h, rank, nranks, unique_id), "comm_init")

    def comm_size(self):
        return LIB.wai_comm_size(self.h)

    def comm_stats(self):
        a, e = C.c_longlong(0), C.c_longlong(0)
        LIB.wai_comm_stats(self.h, C.byref(a), C.byref(e))
        return a.value, e.value

    def mute_comm(self, on):
        """timing probe: collectives return without calling RCCL (every rank together)"""
        self._chk(LIB.wai_bench_mute_comm(self.h, 1 if on else 0), "bench_mute_comm")

    def drop_stream_wait(self, which):
        """fault injection (tests): 1 the face bricks' launch does not wait for the halo exchange, 0 off"""
        self._chk(LIB.wai_test_drop_stream_wait(self.h, int(which)), "test_drop_stream_wait")

    def halo_size(self, dof=None):
        """(bytes this rank sends per halo exchange of a dof-per-cell vector, neighbours)"""
        b, n = C.c_longlong(0), C.c_int(0)
        LIB.wai_halo_size(self.h, self.num_primary_variables if dof is None else dof, C.byref(b), C.byref(n))
        return b.value, n.value

    def launch_stats(self):
        """(kernels launched, copies enqueued) by the linear-solver helpers so far"""
        a, e = C.c_longlong(0), C.c_longlong(0)
        LIB.wai_launch_stats(self.h, C.byref(a), C.byref(e))
        return a.value, e.value

    def pc_kernel_name(self, composed=False):
        """kernel (or path) of a preconditioned-operator application; composed: its form with the operand R - alpha V
        made inside the launch (the second fused launch of a BiCGStab iteration where bcgs_composed() says so)"""
        name = LIB.wai_pc_kernel_name(self.h).decode()
        if composed:
            name = name[:-1] + ",composed>" if name.endswith(">") else name + " (composed operand)"
        return name

    def bcgs_composed(self):
        return LIB.wai_bcgs_composed(self.h) == 1

    def set_regions(self, region):
        r = _lib._i32(region)
        assert r.size == self.n_prim
        self._chk(LIB.wai_set_regions(self.h, r.ctypes.data_as(_lib.pi)), "set_regions")

    def regions(self):
        r = np.zeros(self.n_prim, dtype=np.int32)
        self._chk(LIB.wai_get_regions(self.h, r.ctypes.data_as(_lib.pi)), "get_regions")
        return r

    def fluid(self, which=0):
        """Fluid vector in the reference's AoS layout (n_local, fluid_dof)."""
        out = np.zeros((self.n_local, self.fluid_dof))
        self._chk(LIB.wai_get_fluid(self.h, which, out.ctypes.data), "get_fluid")
        return out

    def fluxes(self):
        """(n_faces, np + nmob): component and phase fluxes per unit area, cell 1 -> cell 2 (the reference's flux vector)"""
        nf = LIB.wai_num_flux_dof(self.h)
        out = np.zeros((self.mesh.n_faces, nf))
        self._chk(LIB.wai_get_fluxes(self.h, out.ctypes.data), "get_fluxes")
        return out

    def source_separated(self):
        """(n_sources, 4): water_rate, water_enthalpy, steam_rate, steam_enthalpy behind each source's separator"""
        out = np.zeros((max(self.mesh.n_src, 1), 4))
        if self.mesh.n_src:
            self._chk(LIB.wai_get_source_separated(self.h, out.ctypes.data_as(_lib.pd)), "get_source_separated")
        return out[: self.mesh.n_src]

    def scale(self, primary, region):
        """eos%scale (src/eos.F90:186-197): unscaled primaries (n, np) -> scaled y."""
        sc = np.ones((9, self.num_primary_variables))
        ps, ts = self.eos_desc.pressure_scale, self.eos_desc.temperature_scale
        for r in (1, 2, 4, 5, 6, 8):     # 5, 6, 8: eos wse regions with halite (src/eos_wse.F90:155-165)
            sc[r, 0] = ps
            if self.num_primary_variables > 1:
                sc[r, 1] = ts if r not in (4, 8) else 1.0
        prim = np.asarray(primary, dtype=np.float64)
        out = prim / sc[np.asarray(region)]
        if self.eos_name in ("wce", "wae") and self.eos_desc.partial_pressure_scale <= 0:
            out[..., 2] = prim[..., 2] / prim[..., 0]  # adaptive Pg / P (eos_wge.F90:639-655)
        if self.eos_name in ("wsce", "wsae"):           # 4th primary: Pg / P, or Pg over its scale
            ppsc = self.eos_desc.partial_pressure_scale
            out[..., 3] = prim[..., 3] / (prim[..., 0] if ppsc <= 0 else ppsc)
        return out

    # ---- ode_type hooks ------------------------------------------------------------------------
    def pre_timestep(self):
        return self._chk(LIB.wai_pre_timestep(self.h), "pre_timestep")

    def pre_try_timestep(self, t):
        return 0  # rock controls (flow_simulation.F90:2039-2089) are out of scope

    def pre_retry_timestep(self):
        return self._chk(LIB.wai_pre_retry_timestep(self.h), "pre_retry_timestep")

    def post_timestep(self):
        return 0

    def pre_iteration(self, y=None):
        return self._chk(LIB.wai_pre_iteration(self.h), "pre_iteration")

    def pre_eval(self, t, y, perturbed_columns=None):
        if perturbed_columns is not None and len(perturbed_columns):
            raise WaiError("coloured perturbation is replaced by wai_jacobian")
        return self._chk(LIB.wai_pre_eval(self.h, t, _lib.ptr(y)), "pre_eval")

    pre_solve = pre_eval

    def lhs(self, t, interval, y, lhs):
        return self._chk(LIB.wai_lhs(self.h, t, _lib.ptr(y), _lib.ptr(lhs)), "lhs")

    def rhs(self, t, interval, y, rhs):
        return self._chk(LIB.wai_rhs(self.h, t, _lib.ptr(y), _lib.ptr(rhs)), "rhs")

    def post_linesearch(self, y_old, search, y):
        cs, cy = C.c_int(0), C.c_int(0)
        err = self._chk(LIB.wai_post_linesearch(self.h, _lib.ptr(y_old), _lib.ptr(search), _lib.ptr(y),
                                                C.byref(cs), C.byref(cy)), "post_linesearch")
        return bool(cs.value), bool(cy.value), err

    def setup_jacobian(self):
        nnzb = LIB.wai_jacobian_nnzb(self.h)
        rp = np.zeros(self.n_owned + 1, dtype=np.int32)
        ci = np.zeros(nnzb, dtype=np.int32)
        self._chk(LIB.wai_jacobian_pattern(self.h, rp.ctypes.data_as(_lib.pi), ci.ctypes.data_as(_lib.pi)),
                  "jacobian_pattern")
        return rp, ci

    # ---- SNES / KSP slots ------------------------------------------------------------------------
    def set_residual_form(self, method="beuler", ratio=0.0, lhs_last2=None):
        """Residual the SNES slots evaluate: the method's `residual` pointer
        (src/timestepper.F90:1484-1500); bdf2 needs ratio = dt / last dt and the lhs two steps back."""
        p = _lib.ptr(lhs_last2) if lhs_last2 is not None else None
        return self._chk(LIB.wai_set_residual_form(self.h, _lib.METHOD_KIND[method], ratio, p),
                         "set_residual_form")

    def set_timestep_method(self, method="beuler"):
        """Method `timestep` integrates with; the library then keeps the BDF2 history."""
        return self._chk(LIB.wai_set_timestep_method(self.h, _lib.METHOD_KIND[method]),
                         "set_timestep_method")

    def residual(self, t, dt, y, lhs_old, f):
        return self._chk(LIB.wai_residual(self.h, t, dt, _lib.ptr(y), _lib.ptr(lhs_old), _lib.ptr(f)), "residual")

    def jacobian(self, t, dt, y, lhs_old):
        return self._chk(LIB.wai_jacobian(self.h, t, dt, _lib.ptr(y), _lib.ptr(lhs_old)), "jacobian")

    def jacobian_values(self):
        nnzb, bs = LIB.wai_jacobian_nnzb(self.h), self.num_primary_variables
        v = np.zeros(nnzb * bs * bs)
        self._chk(LIB.wai_jacobian_get_values(self.h, v.ctypes.data), "jacobian_get_values")
        return v

    def set_jacobian_values(self, val):
        v = _lib._f64(val) if isinstance(val, np.ndarray) else val
        self._chk(LIB.wai_jacobian_set_values(self.h, _lib.ptr(v)), "jacobian_set_values")

    def spmv(self, x, y):
        return self._chk(LIB.wai_spmv(self.h, _lib.ptr(x), _lib.ptr(y)), "spmv")

    def pc_setup(self):
        return self._chk(LIB.wai_pc_setup(self.h), "pc_setup")

    def pc_apply(self, r, z):
        return self._chk(LIB.wai_pc_apply(self.h, _lib.ptr(r), _lib.ptr(z)), "pc_apply")

    def ksp_solve(self, b, x):
        its, reason, rn = C.c_int(0), C.c_int(0), C.c_double(0)
        self._chk(LIB.wai_ksp_solve(self.h, _lib.ptr(b), _lib.ptr(x), C.byref(its), C.byref(reason),
                                    C.byref(rn)), "ksp_solve")
        return its.value, reason.value, rn.value

    def max_scaled(self, v, scale, tol):
        val, idx = C.c_double(0), C.c_int(0)
        self._chk(LIB.wai_max_scaled(self.h, _lib.ptr(v), _lib.ptr(scale), tol, C.byref(val), C.byref(idx)),
                  "max_scaled")
        return val.value, idx.value

    def newton_step(self, t, dt, it, y, lhs_old, f):
        k, r, m = C.c_int(0), C.c_int(0), C.c_double(0)
        self._chk(LIB.wai_newton_step(self.h, t, dt, it, _lib.ptr(y), _lib.ptr(lhs_old), _lib.ptr(f),
                                      C.byref(k), C.byref(r), C.byref(m)), "newton_step")
        return r.value, k.value, m.value

    def timestep(self, t, dt, y):
        n, k, r = C.c_int(0), C.c_int(0), C.c_int(0)
        self._chk(LIB.wai_timestep(self.h, t, dt, _lib.ptr(y), C.byref(n), C.byref(k), C.byref(r)), "timestep")
        return r.value, n.value, k.value

    def set_tracer_injection(self, injection):
        """[n_sources][nt] tracer injection rates (tracer table controls set them per step interval)"""
        self._chk(LIB.wai_set_tracer_injection(self.h, _lib.ptr(_lib._f64(np.asarray(injection, dtype=np.float64)))),
                  "set_tracer_injection")

    # ---- tracers: the auxiliary linear problem (ode_type aux_lhs / aux_rhs / aux_pre_solve) ------
    def set_tracers(self, phase, decay=None, activation=None, diffusion=None, bc=None, injection=None):
        """Passive tracers (src/tracer.F90:30-40): 0-based phase index, decay constant, activation
        energy, diffusion coefficient per tracer; bc [n_bc][nt] Dirichlet mass fractions, injection
        [n_sources][nt] tracer injection rates."""
        nt = len(phase)
        ph = np.ascontiguousarray(phase, dtype=np.int32)

        def arr(a):
            return _lib._f64(np.zeros(nt) if a is None else np.asarray(a, dtype=np.float64))
        dc, ac, df = arr(decay), arr(activation), arr(diffusion)
        self._chk(LIB.wai_set_tracers(self.h, nt, ph.ctypes.data_as(C.POINTER(C.c_int)),
                                      dc.ctypes.data_as(C.POINTER(C.c_double)),
                                      ac.ctypes.data_as(C.POINTER(C.c_double)),
                                      df.ctypes.data_as(C.POINTER(C.c_double))), "set_tracers")
        self.num_tracers = nt
        self.auxiliary = nt > 0
        if bc is not None:
            self._chk(LIB.wai_set_tracer_bc(self.h, _lib.ptr(_lib._f64(np.asarray(bc, dtype=np.float64)))), "set_tracer_bc")
        if injection is not None:
            self._chk(LIB.wai_set_tracer_injection(self.h, _lib.ptr(_lib._f64(np.asarray(injection, dtype=np.float64)))),
                      "set_tracer_injection")

    def set_aux_solver(self, ksp_type="gmres", restart=30, rtol=1e-5, atol=1e-50, max_its=10000):
        return self._chk(LIB.wai_set_aux_solver(self.h, {"bcgs": 0, "gmres": 1}[ksp_type], restart, rtol, atol,
                                                max_its), "set_aux_solver")

    def aux_lhs(self, t, interval, Al):
        return self._chk(LIB.wai_tracer_lhs(self.h, _lib.ptr(Al)), "tracer_lhs")

    def aux_system(self, tracer, method, dt, ratio, alx_last, alx_last2):
        """(scalar CSR values on setup_jacobian()'s pattern, rhs) of one tracer's system"""
        nnzb = LIB.wai_jacobian_nnzb(self.h)
        val, b = np.zeros(nnzb), np.zeros(self.n_owned)
        self._chk(LIB.wai_tracer_system(self.h, tracer, _lib.METHOD_KIND[method], dt, ratio, _lib.ptr(alx_last),
                                        _lib.ptr(alx_last2), val.ctypes.data, b.ctypes.data), "tracer_system")
        return val, b

    def aux_solve(self, method, dt, ratio, alx_last, alx_last2, X, alx_new):
        """setup_linear + aux_pre_solve + KSPSolve (timestepper.F90:2345-2355); (reason, its)"""
        its, reason = C.c_int(0), C.c_int(0)
        self._chk(LIB.wai_tracer_solve(self.h, _lib.METHOD_KIND[method], dt, ratio, _lib.ptr(alx_last),
                                       _lib.ptr(alx_last2), _lib.ptr(X), _lib.ptr(alx_new), C.byref(its),
                                       C.byref(reason)), "tracer_solve")
        return reason.value, its.value

    # ---- measurement ---------------------------------------------------------------------------
    def timer_start(self):
        self._chk(LIB.wai_timer_start(self.h), "timer_start")

    def timer_stop(self):
        ms = C.c_float(0)
        self._chk(LIB.wai_timer_stop(self.h, C.byref(ms)), "timer_stop")
        return ms.value

    def bench_kernel(self, which, reps=100):
        ms = C.c_float(0)
        self._chk(LIB.wai_bench_kernel(self.h, which, reps, C.byref(ms)), "bench_kernel")
        return ms.value

    def synchronize(self):
        self._chk(LIB.wai_synchronize(self.h), "synchronize")

    def profile(self, on=True):
        LIB.wai_profile_enable(self.h, 1 if on else 0)
        LIB.wai_profile_reset(self.h)

    def profile_get(self):
        out = {}
        for k, name in enumerate(_lib.KCLASS):
            ms, n = C.c_double(0), C.c_longlong(0)
            LIB.wai_profile_get(self.h, k, C.byref(ms), C.byref(n))
            out[name] = (ms.value, n.value)
        return out
