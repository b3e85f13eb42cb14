"""ctypes binding of libwaiwera_hip.so (C ABI: include/waiwera_hip.h).

There is no CPU fallback: if the HIP library has not been built, importing this module raises.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libwaiwera_hip.so")

EOS_W, EOS_WE, EOS_WCE, EOS_WSE, EOS_WAE, EOS_WSCE, EOS_WSAE = 0, 1, 2, 3, 4, 5, 6
EOS_KIND = {"w": EOS_W, "we": EOS_WE, "wce": EOS_WCE, "wse": EOS_WSE, "wae": EOS_WAE, "wsce": EOS_WSCE,
            "wsae": EOS_WSAE}
RP = {"fully_mobile": 0, "fully mobile": 0, "linear": 1, "pickens": 2, "corey": 3, "grant": 4,
      "van_genuchten": 5, "van genuchten": 5, "table": 6}
CP = {"zero": 0, "linear": 1, "van_genuchten": 2, "van genuchten": 2, "table": 3}
INTERP = {"linear": 0, "step": 1, "pchip": 2}   # src/interpolation.F90:139-156
KSP = {"bcgs": 0, "gmres": 1, "bcgsl": 2, "lgmres": 3}
PC = {"bjacobi": 0, "asm": 1, "none": 2, "lu": 3}   # linear.preconditioner.type (src/timestepper.F90:1745-1757)
KCLASS = ["eos", "residual", "jacobian", "spmv", "pc_apply", "pc_setup", "vector", "transitions"]

d, i32 = C.c_double, C.c_int
pd, pi = C.POINTER(C.c_double), C.POINTER(C.c_int)


THERMO = {"iapws": 0, "ifc67": 1}   # "thermodynamics" (src/thermodynamics_setup.F90)
METHOD_KIND = {"beuler": 0, "bdf2": 1, "directss": 2}  # src/timestepper.F90:2262-2275


class MeshDesc(C.Structure):
    _fields_ = [("n_owned", i32), ("n_halo", i32), ("n_bc", i32), ("n_faces", i32),
                ("face_cells", pi), ("face_geom", pd), ("cell_geom", pd), ("rock", pd),
                ("n_sub", i32), ("sub_ptr", pi)]


class EosDesc(C.Structure):
    _fields_ = [("kind", i32), ("temperature", d), ("pressure_scale", d), ("temperature_scale", d),
                ("rp_type", i32), ("rp_par", d * 6), ("cp_type", i32), ("cp_par", d * 6),
                ("partial_pressure_scale", d), ("thermo", i32), ("perm_type", i32), ("perm_par", d * 3)]


PERM = {"none": 0, "power": 1, "verma-pruess": 2, "verma_pruess": 2}


class SourceControl(C.Structure):
    """wai_source_control (include/waiwera_hip.h): state-dependent control of one source"""
    _fields_ = [("kind", i32), ("direction", i32), ("limiter", i32), ("table_coord", i32), ("n_table", i32),
                ("coef", d), ("pressure", d), ("limit", d), ("sep_hf", d), ("sep_hg", d), ("table", d * 16),
                ("factor", d), ("sep_more", d * 6), ("threshold", d), ("threshold_pi", d)]


SRC_KIND = {"rate": 0, "deliverability": 1, "recharge": 2}
SRC_DIRECTION = {"both": 0, "production": 1, "out": 1, "injection": 2, "in": 2}
SRC_LIMITER = {None: 0, "total": 1, "water": 2, "steam": 3}


def source_controls(records):
    """array of SourceControl from dicts {kind, direction, limiter, coef, pressure, limit, sep_hf,
    sep_hg, sep_more, table_coord, table} (missing keys: no control of that sort)"""
    arr = (SourceControl * max(len(records), 1))()
    for k, r in zip(arr, records):
        k.kind = SRC_KIND[r.get("kind", "rate")]
        k.direction = SRC_DIRECTION[r.get("direction", "both")]
        k.limiter = SRC_LIMITER[r.get("limiter")]
        k.coef, k.pressure = r.get("coef", 0.0), r.get("pressure", 0.0)
        k.limit, k.sep_hf, k.sep_hg = r.get("limit", 0.0), r.get("sep_hf", 0.0), r.get("sep_hg", 0.0)
        k.factor = r.get("factor", 0.0)
        k.threshold, k.threshold_pi = r.get("threshold", 0.0), r.get("threshold_pi", -1.0)
        more = r.get("sep_more", ())      # (hf, hg) of separator stages 2..4
        if len(more) > 3:
            raise ValueError("separators have at most 4 stages")
        for q, (hf, hg) in enumerate(more):
            k.sep_more[2 * q], k.sep_more[2 * q + 1] = hf, hg
        tab = r.get("table")
        k.table_coord = {None: 0, "enthalpy": 1, "pressure": 2}[r.get("table_coord")] if tab is not None else 0
        if tab is not None:
            if len(tab) > 8:
                raise ValueError("reference pressure tables hold at most 8 points")
            k.n_table = len(tab)
            for q, (x, v) in enumerate(tab):
                k.table[2 * q], k.table[2 * q + 1] = x, v
    return arr


class SolverOpts(C.Structure):
    _fields_ = [("ksp_type", i32), ("gmres_restart", i32), ("ksp_max_its", i32),
                ("ksp_rtol", d), ("ksp_atol", d), ("max_newton_its", i32),
                ("ftol_rel", d), ("ftol_abs", d), ("utol_rel", d), ("utol_abs", d),
                ("fd_eps", d), ("fd_umin", d), ("min_newton_its", i32), ("pc_type", i32), ("asm_overlap", i32),
                ("ilu_levels", i32)]


class WaiError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libwaiwera_hip.so is not built (python -m waiwera_amd.build); the Newton-step hot "
            "path has no CPU fallback")
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    sig = {
        "wai_default_eos": (None, [C.POINTER(EosDesc), i32]),
        "wai_default_opts": (None, [C.POINTER(SolverOpts)]),
        "wai_ctx_create": (i32, [C.POINTER(MeshDesc), C.POINTER(EosDesc), C.POINTER(SolverOpts), i32,
                                 C.POINTER(vp)]),
        "wai_ctx_destroy": (i32, [vp]),
        "wai_last_error": (C.c_char_p, [vp]),
        "wai_set_opts": (i32, [vp, C.POINTER(SolverOpts)]),
        "wai_set_bc": (i32, [vp, pd, pi]),
        "wai_set_curve_table": (i32, [vp, i32, i32, i32, pd]),
        "wai_set_sources": (i32, [vp, i32, pi, pd, pd, pi]),
        "wai_update_sources": (i32, [vp, pd, pd]),
        "wai_update_rock": (i32, [vp, i32, i32, pi, pd]),
        "wai_set_source_controls": (i32, [vp, C.POINTER(SourceControl)]),
        "wai_get_source_rates": (i32, [vp, pd, pd]),
        "wai_set_source_network": (i32, [vp, pi, pi, i32, pi, pi, pi, pi, pi, pd, pd, i32, pi, pi, pi, pi, pi, pi, pd, pd, pd, pi, pi]),
        "wai_network_evaluate": (i32, [i32, pd, pd, pd, pi, pi, i32, pi, pi, pi, pi, pi, pd, pd, i32, pi, pi, pi, pi, pi, pi, pd, pd,
                                       pd, pi, pi, pd, pd, pd]),
        "wai_network_cells": (i32, [i32, pi, pi, pi, i32, pi, pi, pi, pi, pi, pd, pd, i32, pi, pi, pi, pi, pi, pi, pd, pd,
                                    pd, pi, pi, pi, pi]),
        "wai_get_source_network": (i32, [vp, pd, pd]),
        "wai_set_source_global_index": (i32, [vp, i32, pi]),
        "wai_set_network_couplings": (i32, [vp, i32]),
        "wai_get_network_couplings": (i32, [vp, pi, pi, pd]),
        "wai_separator_enthalpies": (i32, [vp, d, pd, pd]),
        "wai_set_regions": (i32, [vp, pi]),
        "wai_get_regions": (i32, [vp, pi]),
        "wai_get_fluid": (i32, [vp, i32, vp]),
        "wai_num_fluid_dof": (i32, [vp]),
        "wai_get_fluxes": (i32, [vp, vp]),
        "wai_num_flux_dof": (i32, [vp]),
        "wai_get_source_separated": (i32, [vp, pd]),
        "wai_block_size": (i32, [vp]),
        "wai_set_halo": (i32, [vp, i32, pi, pi, pi, pi]),
        "wai_comm_unique_id": (i32, [C.c_char_p]),
        "wai_comm_init": (i32, [vp, i32, i32, C.c_char_p]),
        "wai_halo_exchange": (i32, [vp, vp, i32]),
        "wai_comm_size": (i32, [vp]),
        "wai_comm_stats": (i32, [vp, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
        "wai_launch_stats": (i32, [vp, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
        "wai_bench_mute_comm": (i32, [vp, i32]),
        "wai_test_drop_partials": (i32, [vp, i32]),
        "wai_test_drop_stream_wait": (i32, [vp, i32]),
        "wai_halo_size": (i32, [vp, i32, C.POINTER(C.c_longlong), C.POINTER(i32)]),
        "wai_pc_kernel_name": (C.c_char_p, [vp]),
        "wai_pre_timestep": (i32, [vp]),
        "wai_pre_retry_timestep": (i32, [vp]),
        "wai_pre_iteration": (i32, [vp]),
        "wai_pre_eval": (i32, [vp, d, vp]),
        "wai_lhs": (i32, [vp, d, vp, vp]),
        "wai_rhs": (i32, [vp, d, vp, vp]),
        "wai_post_linesearch": (i32, [vp, vp, vp, vp, pi, pi]),
        "wai_set_residual_form": (i32, [vp, i32, d, vp]),
        "wai_set_timestep_method": (i32, [vp, i32]),
        "wai_residual": (i32, [vp, d, d, vp, vp, vp]),
        "wai_jacobian": (i32, [vp, d, d, vp, vp]),
        "wai_jacobian_nnzb": (i32, [vp]),
        "wai_jacobian_pattern": (i32, [vp, pi, pi]),
        "wai_jacobian_get_values": (i32, [vp, vp]),
        "wai_jacobian_set_values": (i32, [vp, vp]),
        "wai_spmv": (i32, [vp, vp, vp]),
        "wai_pc_setup": (i32, [vp]),
        "wai_pc_apply": (i32, [vp, vp, vp]),
        "wai_ksp_solve": (i32, [vp, vp, vp, pi, pi, pd]),
        "wai_max_scaled": (i32, [vp, vp, vp, d, pd, pi]),
        "wai_newton_step": (i32, [vp, d, d, i32, vp, vp, vp, pi, pi, pd]),
        "wai_timestep": (i32, [vp, d, d, vp, pi, pi, pi]),
        "wai_set_tracers": (i32, [vp, i32, pi, pd, pd, pd]),
        "wai_set_tracer_bc": (i32, [vp, vp]),
        "wai_set_tracer_injection": (i32, [vp, vp]),
        "wai_set_aux_solver": (i32, [vp, i32, i32, d, d, i32]),
        "wai_tracer_lhs": (i32, [vp, vp]),
        "wai_tracer_system": (i32, [vp, i32, i32, d, d, vp, vp, vp, vp]),
        "wai_tracer_solve": (i32, [vp, i32, d, d, vp, vp, vp, vp, pi, pi]),
        "wai_timer_start": (i32, [vp]),
        "wai_timer_stop": (i32, [vp, C.POINTER(C.c_float)]),
        "wai_synchronize": (i32, [vp]),
        "wai_bench_kernel": (i32, [vp, i32, i32, C.POINTER(C.c_float)]),
        "wai_bcgs_composed": (i32, [vp]),
        "wai_profile_enable": (i32, [vp, i32]),
        "wai_profile_get": (i32, [vp, i32, pd, C.POINTER(C.c_longlong)]),
        "wai_profile_reset": (i32, [vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    return L, sorted(sig)


LIB, EXPORTED = _load()


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def ptr(a):
    """Raw address of a numpy array (host) or a torch tensor (device or host)."""
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
        return a.ctypes.data
    if hasattr(a, "data_ptr"):
        assert a.is_contiguous()
        return a.data_ptr()
    return int(a)


def default_opts(**kw):
    o = SolverOpts()
    LIB.wai_default_opts(C.byref(o))
    for k, v in kw.items():
        if k == "ksp_type" and isinstance(v, str):
            v = KSP[v]
        if k == "pc_type" and isinstance(v, str):
            v = PC[v]
        setattr(o, k, v)
    return o


def eos_desc(kind="we", temperature=20.0, relperm=("linear", [0.0, 1.0, 0.0, 1.0]),
             capillary=("zero", []), pressure_scale=1.0e6, temperature_scale=1.0e2, thermo="iapws",
             permeability_modifier=None):
    e = EosDesc()
    LIB.wai_default_eos(C.byref(e), EOS_KIND[kind] if isinstance(kind, str) else kind)
    e.temperature = temperature
    e.pressure_scale = pressure_scale
    e.temperature_scale = temperature_scale
    e.rp_type = RP[relperm[0]]
    if relperm[0] != "table":      # ("table", {"liquid": [[S, k], ...], "vapour": ..., "interpolation": ...})
        for k, v in enumerate(relperm[1]):
            e.rp_par[k] = v
    e.cp_type = CP[capillary[0]]
    if capillary[0] != "table":    # ("table", {"pressure": [[S, P], ...], "interpolation": ...})
        for k, v in enumerate(capillary[1]):
            e.cp_par[k] = v
    e.thermo = THERMO[thermo]
    if permeability_modifier is not None:     # (type, [exponent, phir, gamma]), eos wse only
        e.perm_type = PERM[permeability_modifier[0].lower()]
        for k, v in enumerate(permeability_modifier[1]):
            e.perm_par[k] = v
    return e


def comm_unique_id():
    buf = C.create_string_buffer(128)
    if LIB.wai_comm_unique_id(buf) != 0:
        raise WaiError("cannot create an RCCL unique id")
    return buf.raw


def network_arrays(spec):
    """the flat arrays of wai_set_source_network / wai_network_evaluate from a network description:
    dict(rate_specified, enthalpy_specified per source; groups [dict(inputs=[(kind, index)], scaling 0 | 1,
    limits=[(type, limit)], separator=[(hf, hg), ...] or None)]; reinjectors [dict(input=(kind, index),
    outputs=[dict(flow 1 | 2, out=(kind, index), rate, proportion, enthalpy)], overflow=(kind, index))]);
    kinds 0 none, 1 source, 2 group, 3 reinjector.  Returns (arrays kept alive, ctypes arguments)"""
    g, r = spec["groups"], spec["reinjectors"]
    gptr = _i32(np.concatenate([[0], np.cumsum([len(x["inputs"]) for x in g])])) if g else _i32([0])
    gk = _i32([k for x in g for k, _ in x["inputs"]] or [0])
    gi = _i32([i for x in g for _, i in x["inputs"]] or [0])
    gs = _i32([x["scaling"] for x in g] or [0])
    glt, gl = np.full(3 * max(len(g), 1), -1, dtype=np.int32), np.zeros(3 * max(len(g), 1))
    gsep = np.zeros(8 * max(len(g), 1))
    for q, x in enumerate(g):
        for j, (t, v) in enumerate(x["limits"]):
            glt[3 * q + j], gl[3 * q + j] = t, v
        for j, (hf, hg) in enumerate(x.get("separator") or []):
            gsep[8 * q + 2 * j], gsep[8 * q + 2 * j + 1] = hf, hg
    rk = _i32([x["input"][0] for x in r] or [0])
    ri = _i32([x["input"][1] for x in r] or [0])
    rptr = _i32(np.concatenate([[0], np.cumsum([len(x["outputs"]) for x in r])])) if r else _i32([0])
    outs = [o for x in r for o in x["outputs"]]
    of = _i32([o["flow"] for o in outs] or [1])
    ok = _i32([o["out"][0] for o in outs] or [0])
    on = _i32([o["out"][1] for o in outs] or [0])
    orate = _f64([o["rate"] for o in outs] or [0])
    oprop = _f64([o["proportion"] for o in outs] or [0])
    oenth = _f64([o["enthalpy"] for o in outs] or [0])
    vk = _i32([x["overflow"][0] for x in r] or [0])
    vi = _i32([x["overflow"][1] for x in r] or [0])
    rs, es = _i32(spec["rate_specified"]), _i32(spec["enthalpy_specified"])
    keep = [rs, es, gptr, gk, gi, gs, glt, gl, gsep, rk, ri, rptr, of, ok, on, orate, oprop, oenth, vk, vi]
    P = lambda a: a.ctypes.data_as(pi if a.dtype == np.int32 else pd)   # noqa: E731
    args = [P(rs), P(es), len(g), P(gptr), P(gk), P(gi), P(gs), P(glt), P(gl), P(gsep), len(r), P(rk), P(ri), P(rptr),
            P(of), P(ok), P(on), P(orate), P(oprop), P(oenth), P(vk), P(vi)]
    return keep, args


def network_cells(spec, source_cell):
    """the cells between which the network's Jacobian coupling blocks are formed (wai_network_cells)"""
    keep, args = network_arrays(spec)
    sc = _i32(source_cell)
    out = np.zeros(len(sc), dtype=np.int32)
    m = C.c_int(0)
    rc = LIB.wai_network_cells(len(sc), sc.ctypes.data_as(pi), *args, C.byref(m), out.ctypes.data_as(pi))
    if rc != 0:
        raise WaiError("wai_network_cells failed (%d)" % rc)
    return out[: m.value].copy()


def network_evaluate(spec, rate, enthalpy, src_sep=None):
    """one network pass on the host (wai_network_evaluate): node states of sources (n, 6), groups (n, 6),
    reinjectors (n, 8)"""
    n = len(rate)
    keep, args = network_arrays(spec)
    q, h = _f64(rate), _f64(enthalpy)
    sep = _f64(src_sep if src_sep is not None else np.zeros(8 * n))
    S = np.zeros((n, 6)); G = np.zeros((max(len(spec["groups"]), 1), 6)); R = np.zeros((max(len(spec["reinjectors"]), 1), 8))
    rc = LIB.wai_network_evaluate(n, q.ctypes.data_as(pd), h.ctypes.data_as(pd), sep.ctypes.data_as(pd), *args,
                                  S.ctypes.data_as(pd), G.ctypes.data_as(pd), R.ctypes.data_as(pd))
    if rc != 0:
        raise WaiError("wai_network_evaluate failed (%d)" % rc)
    return S, G[: len(spec["groups"])], R[: len(spec["reinjectors"])]
