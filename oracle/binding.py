"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Nothing in waiwera_amd/ may import this module.
"""
import ctypes as C

import numpy as np

d = C.c_double
i32 = C.c_int
pd = C.POINTER(C.c_double)
pi = C.POINTER(C.c_int)


class CurveTable(C.Structure):
    _fields_ = [("n", i32), ("interp", i32), ("x", d * 12), ("v", d * 12), ("d", d * 12)]


class Eos(C.Structure):
    _fields_ = [("kind", i32), ("np", i32), ("nc", i32), ("nph", i32), ("nmob", i32),
                ("df", i32), ("isothermal", i32), ("temperature", d),
                ("scale", d * 4 * 9), ("rp_type", i32), ("cp_type", i32),
                ("rp_par", d * 6), ("cp_par", d * 6), ("thermo", i32), ("perm_type", i32), ("perm_par", d * 3),
                ("tab", CurveTable * 3)]


class NewtonOpts(C.Structure):
    _fields_ = [("ksp_type", i32), ("restart", i32), ("ksp_maxits", i32),
                ("max_newton_its", i32), ("jac_mode", i32),
                ("ksp_rtol", d), ("ksp_atol", d), ("ftol_rel", d), ("ftol_abs", d),
                ("utol_rel", d), ("utol_abs", d), ("fd_eps", d), ("fd_umin", d), ("min_newton_its", i32)]


ROOTFN = C.CFUNCTYPE(d, d, C.c_void_p)
HALOFN = C.CFUNCTYPE(None, C.c_void_p, pd, i32)
ARFN = C.CFUNCTYPE(None, C.c_void_p, pd, i32, i32)

RP = {"fully_mobile": 0, "linear": 1, "pickens": 2, "corey": 3, "grant": 4, "van_genuchten": 5, "table": 6}
CP = {"zero": 0, "linear": 1, "van_genuchten": 2, "table": 3}
INTERP = {"linear": 0, "step": 1, "pchip": 2}


def set_curve_tables(L, eos, relperm=None, capillary=None):
    """("table", {"liquid": [[x, v], ...], "vapour": ..., "interpolation": ...}) / ("table", {"pressure": ...})"""
    if relperm is not None and relperm[0] == "table":
        spec = relperm[1]
        for which, key in ((0, "liquid"), (1, "vapour")):
            xy = f64(spec.get(key, [[0, 0], [1, 1]]))
            assert L.wo_eos_set_curve_table(C.byref(eos), which, INTERP[spec.get("interpolation", "linear")], len(xy), dp(xy)) == 0
    if capillary is not None and capillary[0] == "table":
        spec = capillary[1]
        xy = f64(spec.get("pressure", [[0, 0], [1, 0]]))
        assert L.wo_eos_set_curve_table(C.byref(eos), 2, INTERP[spec.get("interpolation", "linear")], len(xy), dp(xy)) == 0


def dp(a):
    return a.ctypes.data_as(pd)


def ip(a):
    return a.ctypes.data_as(pi)


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def i32a(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def load(path):
    L = C.CDLL(path)
    sig = {
        "wo_region1": (i32, [d, d, pd, pd]), "wo_region2": (i32, [d, d, pd, pd]),
        "wo_sat_pressure": (i32, [d, pd]), "wo_sat_temperature": (i32, [d, pd]),
        "wo_ifc67_region1": (i32, [d, d, d, pd, pd]), "wo_ifc67_region2": (i32, [d, d, pd, pd]),
        "wo_ifc67_sat_pressure": (i32, [d, pd]), "wo_ifc67_sat_temperature": (i32, [d, pd]),
        "wo_ifc67_viscosity": (d, [i32, d, d, d]), "wo_ifc67_phase_composition": (i32, [i32]),
        "wo_viscosity": (d, [d, d]), "wo_phase_composition": (i32, [i32, d, d]),
        "wo_relperm": (None, [i32, pd, d, pd]), "wo_capillary": (d, [i32, pd, d, d]),
        "wo_brent": (i32, [ROOTFN, C.c_void_p, d, d, d, d, i32, pd, pi]),
        "wo_eos_init": (None, [C.POINTER(Eos), i32]),
        "wo_eos_set_curve_table": (i32, [C.POINTER(Eos), i32, i32, i32, pd]),
        "wo_curve_table_value": (d, [C.POINTER(CurveTable), d]),
        "wo_eos_bulk_properties": (i32, [C.POINTER(Eos), pd, pd]),
        "wo_eos_phase_properties": (i32, [C.POINTER(Eos), pd, pd]),
        "wo_eos_transition": (i32, [C.POINTER(Eos), pd, pd, pd, pd, pi]),
        "wo_eos_check_primary": (i32, [C.POINTER(Eos), pd, pd, pi]),
        "wo_co2_properties": (i32, [d, d, pd, pd]), "wo_co2_henrys_constant": (d, [d]),
        "wo_co2_energy_solution": (d, [d]), "wo_co2_viscosity": (i32, [d, d, pd]),
        "wo_ncg_mole_to_mass": (d, [d, d]),
        "wo_eos_scale": (None, [C.POINTER(Eos), pd, i32, pd]),
        "wo_eos_unscale": (None, [C.POINTER(Eos), pd, i32, pd]),
        "wo_cell_balance": (None, [C.POINTER(Eos), pd, pd, pd]),
        "wo_face_flux": (None, [C.POINTER(Eos), pd, pd, pd, pd, pd, pd]),
        "wo_face_phase_density": (d, [C.POINTER(Eos), pd, pd, i32]),
        "wo_conductivity": (d, [pd, pd, C.POINTER(Eos)]),
        "wo_bcsr_spmv": (None, [i32, i32, pi, pi, pd, pd, pd]),
        "wo_bilu0_factor": (i32, [i32, i32, pi, pi, pd, i32, pi, pd, pd]),
        "wo_bilu0_apply": (None, [i32, i32, pi, pi, pd, pd, i32, pi, pd, pd]),
        "wo_sim_create": (C.c_void_p, [i32, i32, i32, i32, i32, pi, pd, pd, pd]),
        "wo_sim_destroy": (None, [C.c_void_p]),
        "wo_sim_eos": (C.POINTER(Eos), [C.c_void_p]),
        "wo_sim_set_comm": (None, [C.c_void_p, HALOFN, ARFN, C.c_void_p]),
        "wo_sim_set_sources": (None, [C.c_void_p, i32, pi, pd, pd, pi]),
        "wo_sim_update_sources": (None, [C.c_void_p, pd, pd]),
        "wo_sim_set_source_controls": (None, [C.c_void_p, C.c_void_p]),
        "wo_sim_source_rates": (None, [C.c_void_p, pd, pd]),
        "wo_separator_enthalpies": (i32, [C.c_void_p, C.c_double, pd, pd]),
        "wo_separator_steam_fraction": (C.c_double, [C.c_void_p, C.c_double]),
        "wo_permeability_factor": (d, [C.c_void_p, d]),
        "wo_gas_henry_salt": (None, [C.c_void_p, d, d, pd, pd]), "wo_co2_henrys_constant": (d, [d]),
        "wo_air_properties": (i32, [d, d, pd, pd]), "wo_air_henrys_constant": (d, [d]),
        "wo_air_energy_solution": (d, [d]), "wo_air_mixture_viscosity": (d, [d, d, d]),
        "wo_halite_solubility": (i32, [d, pd]), "wo_halite_properties": (i32, [d, d, pd, pd]),
        "wo_halite_solubility_two_phase": (i32, [C.c_void_p, d, pd]),
        "wo_brine_sat_pressure": (i32, [C.c_void_p, d, d, pd]), "wo_brine_sat_temperature": (i32, [C.c_void_p, d, d, pd]),
        "wo_brine_properties": (i32, [C.c_void_p, d, d, d, pd, pd]), "wo_brine_viscosity": (i32, [C.c_void_p, d, d, d, pd]),
        "wo_halite_solubility": (i32, [d, pd]), "wo_halite_properties": (i32, [d, d, pd, pd]),
        "wo_halite_solubility_two_phase": (i32, [C.c_void_p, d, pd]),
        "wo_brine_sat_pressure": (i32, [C.c_void_p, d, d, pd]), "wo_brine_sat_temperature": (i32, [C.c_void_p, d, d, pd]),
        "wo_brine_properties": (i32, [C.c_void_p, d, d, d, pd, pd]), "wo_brine_viscosity": (i32, [C.c_void_p, d, d, d, pd]),
        "wo_sim_set_subdomains": (None, [C.c_void_p, i32, pi]),
        "wo_sim_set_asm": (None, [C.c_void_p, i32]),
        "wo_sim_asm_rows": (i32, [C.c_void_p, pi, pi]),
        "wo_sim_set_pc_none": (None, [C.c_void_p, i32]),
        "wo_sim_set_ilu_levels": (None, [C.c_void_p, i32]),
        "wo_sim_local_pattern": (i32, [C.c_void_p, i32, pi, pi]),
        "wo_sim_spread_pages": (None, [C.c_void_p]),
        "wo_pc_setup": (i32, [C.c_void_p, pd]),
        "wo_pc_apply": (None, [C.c_void_p, pd, pd]),
        "wo_sim_set_regions": (None, [C.c_void_p, pi]),
        "wo_sim_get_regions": (None, [C.c_void_p, pi]),
        "wo_sim_init_bc": (i32, [C.c_void_p, pd, pi]),
        "wo_sim_fluid": (pd, [C.c_void_p]),
        "wo_sim_nnzb": (i32, [C.c_void_p]),
        "wo_sim_pattern": (None, [C.c_void_p, pi, pi]),
        "wo_pre_timestep": (None, [C.c_void_p]), "wo_pre_retry_timestep": (None, [C.c_void_p]),
        "wo_pre_iteration": (None, [C.c_void_p]),
        "wo_pre_eval": (i32, [C.c_void_p, pd]),
        "wo_lhs": (None, [C.c_void_p, pd]), "wo_rhs": (None, [C.c_void_p, pd]),
        "wo_sim_set_residual_form": (i32, [C.c_void_p, i32, d, pd]),
        "wo_sim_set_timestep_method": (i32, [C.c_void_p, i32]),
        "wo_residual": (i32, [C.c_void_p, pd, d, pd, pd]),
        "wo_post_linesearch": (i32, [C.c_void_p, pd, pd, pd, pi, pi]),
        "wo_jacobian": (i32, [C.c_void_p, pd, d, pd, pd, i32, pd]),
        "wo_max_scaled": (None, [C.c_void_p, pd, pd, d, pd, pi]),
        "wo_ksp_solve": (i32, [C.c_void_p, i32, i32, pd, pd, pd, d, d, i32, pi, pd, pd]),
        "wo_newton_opts_default": (None, [C.POINTER(NewtonOpts)]),
        "wo_newton_step": (i32, [C.c_void_p, C.POINTER(NewtonOpts), i32, d, pd, pd, pd, pi, pd]),
        "wo_timestep": (i32, [C.c_void_p, C.POINTER(NewtonOpts), d, pd, pi]),
        "wo_sim_set_tracers": (i32, [C.c_void_p, i32, pi, pd, pd, pd]),
        "wo_sim_set_tracer_bc": (None, [C.c_void_p, pd]),
        "wo_sim_set_tracer_injection": (None, [C.c_void_p, pd]),
        "wo_tracer_lhs": (None, [C.c_void_p, pd]),
        "wo_tracer_system": (None, [C.c_void_p, i32, i32, d, d, pd, pd, pd, pd]),
        "wo_tracer_solve": (i32, [C.c_void_p, i32, d, d, pd, pd, pd, pd, i32, i32, d, d, i32, pi]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    return L


class OracleSim:
    """Thin object wrapper over the oracle's wo_sim for the tests / cpu baseline."""

    def __init__(self, L, mesh, eos_kind, thermo=0, relperm=None, capillary=None, permeability_modifier=None):
        self.L = L
        self.mesh = mesh
        self._keep = [f64(mesh.face_geom), f64(mesh.cell_geom), f64(mesh.rock), i32a(mesh.face_cells)]
        fg, cg, rk, fc = self._keep
        self.h = L.wo_sim_create(eos_kind, mesh.n_owned, mesh.n_halo, mesh.n_bc, mesh.n_faces,
                                 ip(fc), dp(fg), dp(cg), dp(rk))
        self.eos = L.wo_sim_eos(self.h).contents
        self.eos.thermo = thermo   # 0 IAPWS-97, 1 IFC-67; before the boundary fluid is evaluated
        if relperm is not None:    # (type name, parameters) like waiwera_amd.lib.eos_desc
            self.eos.rp_type = RP[relperm[0]]
            if relperm[0] != "table":
                for k, v in enumerate(relperm[1]):
                    self.eos.rp_par[k] = v
        if permeability_modifier is not None:
            self.eos.perm_type = {"power": 1, "verma-pruess": 2}[permeability_modifier[0]]
            for k, v in enumerate(permeability_modifier[1]):
                self.eos.perm_par[k] = v
        if capillary is not None:
            self.eos.cp_type = CP[capillary[0]]
            if capillary[0] != "table":
                for k, v in enumerate(capillary[1]):
                    self.eos.cp_par[k] = v
        set_curve_tables(L, self.eos, relperm, capillary)
        self.np = self.eos.np
        self.df = self.eos.df
        self.n_owned, self.n_prim = mesh.n_owned, mesh.n_owned + mesh.n_halo
        self.n_local = self.n_prim + mesh.n_bc
        if mesh.n_bc:
            bp, br = f64(mesh.bc_primary), i32a(mesh.bc_region)
            assert L.wo_sim_init_bc(self.h, dp(bp), ip(br)) == 0
        if getattr(mesh, "n_src", 0):
            sc, sr, se, sk = i32a(mesh.src_cell), f64(mesh.src_rate), f64(mesh.src_enthalpy), i32a(mesh.src_component)
            L.wo_sim_set_sources(self.h, len(sc), ip(sc), dp(sr), dp(se), ip(sk))
        if getattr(mesh, "sub_ptr", None) is not None:
            sp = i32a(mesh.sub_ptr)
            L.wo_sim_set_subdomains(self.h, len(sp) - 1, ip(sp))
        self._cb = None

    def close(self):
        if self.h:
            self.L.wo_sim_destroy(self.h)
            self.h = None

    def set_source_rates(self, rate=None, enthalpy=None):
        r = f64(rate) if rate is not None else None
        e = f64(enthalpy) if enthalpy is not None else None
        self.L.wo_sim_update_sources(self.h, dp(r) if r is not None else None, dp(e) if e is not None else None)

    def set_source_controls(self, records):
        # wo_src_ctl has the layout of wai_source_control: reuse the host-side packer
        from waiwera_amd.lib import source_controls
        self._ctl = source_controls(records) if records is not None else None
        self.L.wo_sim_set_source_controls(self.h, C.cast(self._ctl, C.c_void_p) if self._ctl is not None else None)

    def separator_enthalpies(self, pressure):
        hf, hg = np.zeros(1), np.zeros(1)
        assert self.L.wo_separator_enthalpies(C.byref(self.eos), pressure, dp(hf), dp(hg)) == 0
        return float(hf[0]), float(hg[0])

    def source_rates(self):
        n = getattr(self.mesh, "n_src", 0)
        r, e = np.zeros(n), np.zeros(n)
        if n:
            self.L.wo_sim_source_rates(self.h, dp(r), dp(e))
        return r, e

    def set_asm(self, overlap, *_):
        """PCASM (restricted) with `overlap` layers around the subdomains; 0: block Jacobi"""
        self.L.wo_sim_set_asm(self.h, int(overlap))

    def set_ilu_levels(self, levels):
        """ILU(k) sub-preconditioner: levels of fill (0: ILU(0)); with block Jacobi or PCASM"""
        self.L.wo_sim_set_ilu_levels(self.h, int(levels))

    def local_pattern(self, sd):
        """(rowptr, colidx) of subdomain sd's local system after fill, local column indices"""
        nz = self.L.wo_sim_local_pattern(self.h, int(sd), None, None)
        ptr, rows = self.asm_rows()
        m = int(ptr[sd + 1] - ptr[sd])
        rp, ci = np.zeros(m + 1, dtype=np.int32), np.zeros(max(nz, 1), dtype=np.int32)
        self.L.wo_sim_local_pattern(self.h, int(sd), ip(rp), ip(ci))
        return rp, ci[:nz]

    def asm_rows(self):
        n = self.L.wo_sim_asm_rows(self.h, None, None)
        ptr = np.zeros(len(self.mesh.sub_ptr), dtype=np.int32)
        rows = np.zeros(n, dtype=np.int32)
        if n:
            self.L.wo_sim_asm_rows(self.h, ip(ptr), ip(rows))
        return ptr, rows

    def pc_setup(self, val):
        return self.L.wo_pc_setup(self.h, dp(f64(val)))

    def pc_apply(self, r):
        z = np.zeros(self.n_owned * self.np)
        self.L.wo_pc_apply(self.h, dp(f64(r)), dp(z))
        return z

    def set_regions(self, region):
        r = i32a(region)
        assert r.size == self.n_prim
        self.L.wo_sim_set_regions(self.h, ip(r))

    def regions(self):
        r = np.zeros(self.n_prim, dtype=np.int32)
        self.L.wo_sim_get_regions(self.h, ip(r))
        return r

    def fluid(self):
        ptr = self.L.wo_sim_fluid(self.h)
        return np.ctypeslib.as_array(ptr, shape=(self.n_local, self.df))

    def pattern(self):
        nnzb = self.L.wo_sim_nnzb(self.h)
        rp = np.zeros(self.n_owned + 1, dtype=np.int32)
        ci = np.zeros(nnzb, dtype=np.int32)
        self.L.wo_sim_pattern(self.h, ip(rp), ip(ci))
        return rp, ci

    def yvec(self, y):
        out = np.zeros(self.n_prim * self.np)
        y = np.asarray(y, dtype=np.float64).ravel()
        out[: y.size] = y
        return out

    def pre_eval(self, y):
        return self.L.wo_pre_eval(self.h, dp(y))

    def lhs(self):
        out = np.zeros(self.n_owned * self.np)
        self.L.wo_lhs(self.h, dp(out))
        return out

    def rhs(self):
        out = np.zeros(self.n_owned * self.np)
        self.L.wo_rhs(self.h, dp(out))
        return out

    def set_residual_form(self, method=0, ratio=0.0, lhs_last2=None):
        p = dp(f64(lhs_last2)) if lhs_last2 is not None else None
        assert self.L.wo_sim_set_residual_form(self.h, method, ratio, p) == 0

    def set_timestep_method(self, method=0):
        assert self.L.wo_sim_set_timestep_method(self.h, method) == 0

    def residual(self, y, dt, lhs_old):
        f = np.zeros(self.n_owned * self.np)
        err = self.L.wo_residual(self.h, dp(y), dt, dp(f64(lhs_old)), dp(f))
        return err, f

    def jacobian(self, y, dt, lhs_old, f, mode=0):
        nnzb = self.L.wo_sim_nnzb(self.h)
        val = np.zeros(nnzb * self.np * self.np)
        err = self.L.wo_jacobian(self.h, dp(y), dt, dp(f64(lhs_old)), dp(f64(f)), mode, dp(val))
        return err, val

    def ksp_solve(self, val, b, ksp_type=0, restart=30, rtol=1e-5, atol=1e-50, maxits=10000):
        x = np.zeros(self.n_prim * self.np)
        its = C.c_int(0)
        rn = C.c_double(0)
        hist = np.zeros(maxits + 2)
        reason = self.L.wo_ksp_solve(self.h, ksp_type, restart, dp(f64(val)), dp(f64(b)), dp(x),
                                     rtol, atol, maxits, C.byref(its), C.byref(rn), dp(hist))
        return reason, x[: self.n_owned * self.np], its.value, hist[: its.value + 1]

    # ---- tracers (auxiliary linear problem) ----------------------------------------------------
    def set_tracers(self, phase, decay=None, activation=None, diffusion=None, bc=None, injection=None):
        nt = len(phase)
        z = np.zeros(nt)
        ph = i32a(phase)
        assert self.L.wo_sim_set_tracers(self.h, nt, ip(ph), dp(f64(decay if decay is not None else z)),
                                         dp(f64(activation if activation is not None else z)),
                                         dp(f64(diffusion if diffusion is not None else z))) == 0
        self.nt = nt
        if bc is not None and self.mesh.n_bc:
            self.L.wo_sim_set_tracer_bc(self.h, dp(f64(bc)))
        if injection is not None and getattr(self.mesh, "n_src", 0):
            self.L.wo_sim_set_tracer_injection(self.h, dp(f64(injection)))

    def set_tracer_injection(self, injection):
        self.L.wo_sim_set_tracer_injection(self.h, dp(f64(injection)))

    def tracer_lhs(self):
        out = np.zeros(self.n_owned * self.nt)
        self.L.wo_tracer_lhs(self.h, dp(out))
        return out

    def tracer_system(self, it, method, dt, ratio, alx_last, alx_last2):
        A, b = np.zeros(self.L.wo_sim_nnzb(self.h)), np.zeros(self.n_owned)
        a2 = f64(alx_last2) if alx_last2 is not None else np.zeros(self.n_owned * self.nt)
        self.L.wo_tracer_system(self.h, it, method, dt, ratio, dp(f64(alx_last)), dp(a2), dp(A), dp(b))
        return A, b

    def tracer_solve(self, method, dt, ratio, alx_last, alx_last2, X, ksp_type=1, restart=30,
                     rtol=1e-5, atol=1e-50, maxits=10000):
        """X: [cell][tracer] in/out (n_owned*nt); returns (reason, its, alx_new)"""
        alx_new = np.zeros(self.n_owned * self.nt)
        its = C.c_int(0)
        a2 = dp(f64(alx_last2)) if alx_last2 is not None else dp(np.zeros(self.n_owned * self.nt))
        r = self.L.wo_tracer_solve(self.h, method, dt, ratio, dp(f64(alx_last)), a2, dp(X), dp(alx_new),
                                   ksp_type, restart, rtol, atol, maxits, C.byref(its))
        return r, its.value, alx_new

    def opts(self):
        o = NewtonOpts()
        self.L.wo_newton_opts_default(C.byref(o))
        return o

    def timestep(self, y, dt, opts=None):
        o = opts or self.opts()
        k = C.c_int(0)
        r = self.L.wo_timestep(self.h, C.byref(o), dt, dp(y), C.byref(k))
        return r, k.value


def jacobian_parity(Jg, Jo, rowptr, colidx, y, lhs, bs, fd_eps=1.0e-8, fd_umin=1.0e-2, bar=False, audit=None):
    """Two finite-difference Jacobians of the same residual function, entry by entry (BCSR values, row-major blocks).

    An entry J[i,r; j,k] = (f_ir(y + h_jk e_jk) - f_ir(y)) / h_jk carries the rounding of f_ir divided by the step:
    two correct evaluations of f that differ in the last bits of its accumulation term L_ir differ in the entry by a
    few eps |L_ir| / |h_jk| -- which for a small scaled primary (a gas partial-pressure fraction of 0.02 has
    h = 2e-10) is 1e-5 of the entry scale.  Returns (worst difference relative to the largest entry of the block
    row's equation, worst difference in units of eps |L_ir| / |h_jk| among the entries above 2e-5 of that scale);
    with bar=True a third value: the worst difference over THE bar an entry has to meet,
    max(2e-5 x the block row's largest entry, 16 eps |L_ir| / |h_jk|) -- parity holds iff it is <= 1.
    `audit` (a dict, filled in): how many entries there are, how many differ by more than 2e-5 of their row's scale (the
    ones only the ulp-step allowance admits), and the largest of those -- so that a drift from "a handful of
    partial-pressure columns" to "most of the matrix" is seen."""
    n = rowptr.size - 1
    Jg = np.asarray(Jg).reshape(-1, bs, bs)
    Jo = np.asarray(Jo).reshape(-1, bs, bs)
    rows = np.repeat(np.arange(n), np.diff(rowptr))
    yb = np.asarray(y, dtype=np.float64)[: (int(colidx.max()) + 1) * bs].reshape(-1, bs)
    dx = np.where(np.abs(yb) < fd_umin, np.where(yb >= 0.0, fd_umin, -fd_umin), yb)
    h = np.abs(dx * fd_eps)                      # MatFDColoring "ds" increment on the scaled variables
    Lb = np.abs(np.asarray(lhs, dtype=np.float64)[: n * bs].reshape(-1, bs))
    eps = np.finfo(np.float64).eps
    worst_rel, worst_ulp, worst_bar = 0.0, 0.0, 0.0
    n_above, largest_above = 0, 0.0
    for r in range(bs):
        rowscale = np.zeros(n)
        np.maximum.at(rowscale, rows, np.abs(Jo[:, r, :]).max(axis=1))
        for k in range(bs):
            d = np.abs(Jg[:, r, k] - Jo[:, r, k])
            rel = d / np.maximum(rowscale[rows], 1e-300)
            worst_rel = max(worst_rel, float(rel.max()))
            big = rel > 2.0e-5
            n_above += int(big.sum())
            if big.any():
                largest_above = max(largest_above, float(rel[big].max()))
                ulp = d[big] / (eps * np.maximum(Lb[rows[big], r], 1e-300) / h[colidx[big], k])
                worst_ulp = max(worst_ulp, float(ulp.max()))
            if bar:
                allow = np.maximum(2.0e-5 * rowscale[rows], 16.0 * eps * np.maximum(Lb[rows, r], 1e-300) / h[colidx, k])
                worst_bar = max(worst_bar, float((d / np.maximum(allow, 1e-300)).max()))
    if audit is not None:
        audit.update({"entries": int(Jg.shape[0] * bs * bs), "entries_above_2e-5_of_row_scale": n_above,
                      "largest_of_them": largest_above})
    return (worst_rel, worst_ulp, worst_bar) if bar else (worst_rel, worst_ulp)
