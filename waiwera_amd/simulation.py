"""Runs a simulation described by a Waiwera JSON input file (the subset the hot path covers) on the
HIP library: the reference's own input format on one side of the path, its output field names on
the other.

Input keys handled, with the reference's defaults (src/flow_simulation.F90:296-670, 800-846;
src/mesh.F90:270-300, 1640-1800; src/rock_setup.F90; src/initial.F90:421-677; src/source_setup.F90;
src/timestepper.F90:1960-2275; src/tracer.F90:63-140; utils/input_schema.json):

  mesh        filename (gmsh MSH 2.2), thickness, radial, zones (all / box ranges on x, y, z)
  gravity     number | vector | null
  eos         name w | we | wce | wae | wse | wsce | wsae, temperature, permeability_modifier
  thermodynamics  iapws | ifc67
  rock        types [cells | zones, permeability, porosity, density, specific_heat, wet / dry
              conductivity], relative_permeability, capillary_pressure
  initial     primary (one record or one per cell), region (one or per cell), tracer; or
              filename + index: restart from a Waiwera HDF5 output file
  boundaries  primary, region, faces {cells, normal} (one or a list), tracer
  mesh        filename (gmsh 2.2 ASCII; or mesh_file=: a MULgraph geometry file), thickness, radial,
              zones (boxes, cell lists), minc {geometry, rock {fracture, matrix, zones | types}}
  source      cell, rate, enthalpy, tracer (constants or [[t, v], ...] tables with "interpolation":
              linear|step and "averaging": integrate|endpoint), component, deliverability {productivity,
              pressure}, recharge | injectivity {coefficient, pressure}, limiter {type, limit,
              separator_pressure}, separator {pressure}, direction, factor
  time        start, stop, step {size, adapt, maximum, method, solver.nonlinear, solver.linear}
  tracer      name, phase, decay, activation, diffusion

  output      filename, initial, final, frequency, checkpoint {time, tolerance} (cell fields, source
              rate and enthalpy)

Anything else that changes results (deliverability thresholds, ...) raises
NotImplementedError instead of being ignored.  Output: `Simulation.run` returns the final cell
fields under the reference's HDF5 dataset names (fluid_pressure, ...) and writes "output.filename"
in the reference's HDF5 layout (waiwera_amd/hdf5io.py, HDF5 C library through ctypes); `save`
writes a .npz archive.
"""
import json
import os

import numpy as np

from . import gmsh, unstructured
from .timestepper import Timestepper
from .interpolation import Table

UNSUPPORTED_SOURCE_KEYS = ()


def _get(d, path, default=None):
    for k in path.split("."):
        if not isinstance(d, dict) or k not in d or d[k] is None:
            return default
        d = d[k]
    return d


def relperm_spec(rp):
    """rock.relative_permeability -> (type, parameters) of the library's EOS descriptor"""
    if rp is None:
        return "linear", [0.0, 1.0, 0.0, 1.0]
    t = rp.get("type", "linear").lower()
    if t == "linear":
        return "linear", list(rp.get("liquid", [0.0, 1.0])) + list(rp.get("vapour", [0.0, 1.0]))
    if t in ("fully mobile", "fully_mobile"):
        return "fully_mobile", []
    if t in ("corey", "grant"):
        return t, [rp.get("slr", 0.3), rp.get("ssr", 0.05 if t == "corey" else 0.6)]
    if t == "pickens":
        return "pickens", [rp.get("power", 1.0)]
    if t in ("van genuchten", "van_genuchten"):
        return "van_genuchten", [rp.get("lambda", 0.45), rp.get("slr", 1.0e-3), rp.get("sls", 1.0),
                                 1.0 if rp.get("sum_unity", True) else 0.0, rp.get("ssr", 0.6)]
    if t == "table":   # src/relative_permeability.F90:500-543
        spec = {"interpolation": rp.get("interpolation", "linear")}
        for k in ("liquid", "vapour"):
            if rp.get(k) is not None:
                spec[k] = [list(map(float, row)) for row in rp[k]]
        return "table", spec
    raise NotImplementedError("relative permeability type %r" % t)


def capillary_spec(cp):
    if cp is None:
        return "zero", []
    t = cp.get("type", "zero").lower()
    if t == "zero":
        return "zero", []
    if t == "linear":
        s = cp.get("saturation_limits", [0.0, 1.0])
        if cp.get("pressure", 0.125e5) == 0:
            return "zero", []
        return "linear", [s[0], s[1], cp.get("pressure", 0.125e5)]
    if t in ("van genuchten", "van_genuchten"):
        pmax = cp.get("Pmax")
        return "van_genuchten", [cp.get("P0", 0.125e5), cp.get("lambda", 0.45), cp.get("slr", 1.0e-3),
                                 cp.get("sls", 1.0), pmax if pmax is not None else 0.0, 1.0 if pmax is not None else 0.0]
    if t == "table":   # src/capillary_pressure.F90:311-345
        spec = {"interpolation": cp.get("interpolation", "linear")}
        if cp.get("pressure") is not None:
            spec["pressure"] = [list(map(float, row)) for row in cp["pressure"]]
        return "table", spec
    raise NotImplementedError("capillary pressure type %r" % t)


def _is_table(v):
    """a rank-2 array: rows of [time, value(s)] (rock_setup.F90:300-326: permeability of rank 2, porosity any array)"""
    return isinstance(v, (list, tuple)) and len(v) > 0 and isinstance(v[0], (list, tuple))


def rock_record(rt, dim):
    k = rt.get("permeability", 1.0e-13)
    if _is_table(k):       # permeability against time: a rock control sets it before every try; start from the first row
        k = list(k[0][1:])
    por = rt.get("porosity", 0.1)
    if isinstance(por, (list, tuple)):
        rt = dict(rt, porosity=(por[0][1] if _is_table(por) else por[0]))
    k = [k] * 3 if np.isscalar(k) else list(k) + [k[-1]] * (3 - len(k))
    return np.array([k[0], k[1], k[2], rt.get("wet_conductivity", 2.5), rt.get("dry_conductivity", rt.get("wet_conductivity", 2.5)),
                     rt.get("porosity", 0.1), rt.get("density", 2200.0), rt.get("specific_heat", 1000.0)])


def zone_cells(zone, centroids):
    """cells of a "mesh.zones" entry: box ranges on x / y / z (a box without ranges is everything)"""
    if zone is None:
        return np.arange(len(centroids))
    if isinstance(zone, dict):
        if "cells" in zone:
            return np.asarray(zone["cells"], dtype=int)
        sel = np.ones(len(centroids), dtype=bool)
        for ax, name in enumerate("xyz"):
            if name in zone and zone[name] is not None:
                lo, hi = zone[name]
                sel &= (centroids[:, ax] >= lo) & (centroids[:, ax] <= hi)
        unknown = set(zone) - {"x", "y", "z", "type", "cells"}
        if unknown:
            raise NotImplementedError("mesh zone keys %s" % sorted(unknown))
        return np.nonzero(sel)[0]
    raise NotImplementedError("mesh zone %r" % (zone,))


def network_spec(inp, interval, separator_enthalpies=None):
    """"network": {"group": [...], "reinject": [...]} of an input file (setup_source_network_groups /
    _reinjectors, src/source_setup.F90) as the description waiwera_amd.lib.network_arrays flattens for
    wai_set_source_network.  Rates, proportions and enthalpies given as time tables are averaged over
    `interval` (table_object_control_update, src/control.F90:263-284).  Returns (spec or None, names of
    the groups / reinjectors in spec order, whether anything is a time table)."""
    net = inp.get("network") or {}
    groups, reinj = list(net.get("group") or []), list(net.get("reinject") or [])
    names = {"group": [g.get("name", "") for g in groups], "reinject": [r.get("name", "") for r in reinj]}
    if not groups and not reinj:
        return None, names, False
    sources = inp.get("source", []) or []
    sidx = {s["name"]: i for i, s in enumerate(sources) if "name" in s}
    gnames = [g.get("name", "") for g in groups]
    order, done = [], set()
    while len(order) < len(groups):      # groups in dependency order
        progress = False
        for gi, g in enumerate(groups):
            if gi in done:
                continue
            if all(n in sidx or (n in gnames and gnames.index(n) in done) for n in g.get("in", [])):
                order.append(gi); done.add(gi); progress = True
        if not progress:
            raise ValueError("network groups refer to unknown nodes or to each other in a cycle")
    groups = [groups[gi] for gi in order]
    gidx = {g.get("name", ""): k for k, g in enumerate(groups)}
    ridx = {r.get("name", ""): k for k, r in enumerate(reinj)}
    names["group"] = [g.get("name", "") for g in groups]
    timed = [False]

    def ref(name, allowed):
        for kind, table in ((1, sidx), (2, gidx), (3, ridx)):
            if kind in allowed and name in table:
                return kind, table[name]
        raise ValueError("network node %r not found" % (name,))

    def value(v, o, default):
        if v is None:
            return default
        if isinstance(v, dict):
            v = v.get("time")
        if isinstance(v, (list, tuple)):
            timed[0] = True
            return float(Table(v, o.get("interpolation", "linear"), o.get("averaging", "integrate")).average(interval)[0])
        return float(v)

    def separator(sp):
        if not sp:
            return None
        if separator_enthalpies is None:
            raise NotImplementedError("separator on a network group needs the thermodynamics (separator_enthalpies)")
        ps = sp.get("pressure", 0.55e6) if isinstance(sp, dict) else 0.55e6
        ps = list(ps) if isinstance(ps, (list, tuple)) else [ps]
        if len(ps) > 4:
            raise NotImplementedError("separators with more than 4 stages")
        return [separator_enthalpies(float(q)) for q in ps]
    FLOW = {"total": 0, "water": 1, "steam": 2}
    # rate_specified: a "rate" in the input (source_setup.F90:2087-2118, get_initial_rate) -- or a deliverability
    # (:2911) or recharge / injectivity (:3081) control, whose set-up sets source%rate_specified = PETSC_TRUE; the
    # control's rate then becomes the specified rate with every source%set_rate (source.F90:276-288), so a
    # reinjector output into such a well is capped by what the control gives (source.F90:292-303)
    spec = dict(rate_specified=[int("rate" in s or any(k in s for k in ("deliverability", "recharge", "injectivity")))
                                for s in sources],
                enthalpy_specified=[int("enthalpy" in s) for s in sources], groups=[], reinjectors=[])
    for g in groups:
        lim = g.get("limiter") or {}
        if "type" in lim:
            lim = {lim["type"]: lim.get("limit")}
        limits = [(FLOW[k], value(lim[k], lim, 0.0)) for k in ("total", "water", "steam") if lim.get(k) is not None]
        spec["groups"].append(dict(inputs=[ref(n, (1, 2)) for n in g.get("in", [])],
                                   scaling={"uniform": 0, "progressive": 1}[g.get("scaling", "uniform")], limits=limits,
                                   separator=separator(g.get("separator"))))
    for r in reinj:
        outs = []
        for flow, key in ((1, "water"), (2, "steam")):
            for o in r.get(key) or []:
                has_rate = o.get("rate") is not None
                outs.append(dict(flow=flow, out=ref(o["out"], (1, 3)) if o.get("out") else (0, -1),
                                 rate=value(o.get("rate"), o, -1.0),
                                 proportion=value(o.get("proportion"), o, -1.0) if not has_rate else -1.0,
                                 enthalpy=value(o.get("enthalpy"), o, -1.0)))
        over = r.get("overflow")
        if isinstance(over, dict):
            over = over.get("out")
        spec["reinjectors"].append(dict(input=ref(r["in"], (1, 2)) if r.get("in") else (0, -1), outputs=outs,
                                        overflow=ref(over, (1, 3)) if over else (0, -1)))
    return spec, names, timed[0]


class Simulation:
    """One Waiwera input file -> mesh, flow simulation object and time stepper."""

    def __init__(self, inp, base_dir=".", ode_factory=None, device=0, mesh_builder=None, mesh_file=None,
                 output_dir=None, rank=0, world=1, comm_id=None, owner=None, default_pc="asm"):
        """rank / world / comm_id (wai_comm_unique_id of rank 0, handed round by the host): one process per rank, each
        reads the whole input, keeps its own cells with one ghost layer (waiwera_amd.partition.partition_mesh; owner: rank of
        every cell, default contiguous blocks of the input's numbering) and runs the same step sequence -- what
        DMPlexDistribute (src/mesh.F90:143-171) does for the reference.  On several ranks: constant or tabulated sources
        and their device-side controls; not (yet) MINC zones, source networks, tracers, rock table controls, output files
        (fields() returns the rank's cells, self.owned_gid their index in the input's numbering)."""
        self.rank, self.world = int(rank), int(world)
        if default_pc not in ("asm", "bjacobi"):
            raise ValueError("default_pc 'asm' (the reference's default) or 'bjacobi' (the library's fused path)")
        self.default_pc, self.pc_choice = default_pc, None
        # output files go to output_dir (default: $WAIWERA_OUTPUT_DIR, else beside the input file)
        self.output_dir = output_dir or os.environ.get("WAIWERA_OUTPUT_DIR") or base_dir
        self.output_error = None
        self.inp = inp
        self.base_dir = base_dir
        mesh = inp.get("mesh")
        if isinstance(mesh, str):
            mesh = {"filename": mesh}
        if mesh_builder is None:
            # gmsh 2.2 ASCII, or (mesh_file: the input names an ExodusII file, which cannot be read
            # here) the MULgraph geometry file the reference's benchmarks keep beside it
            mpath = mesh_file or os.path.join(base_dir, mesh["filename"])
            if mpath.endswith(".dat"):
                from . import mulgrid
                nodes, cells, dim = mulgrid.read_geometry(mpath)
            else:
                nodes, cells, dim = gmsh.read_msh(mpath)
            n = len(cells)
        else:
            # mesh_builder(boundaries, sources) -> LocalMesh for meshes that are not gmsh files (the
            # ExodusII meshes of some reference benchmarks are rebuilt from their description);
            # mesh_builder.dim, mesh_builder.n_cells describe it
            dim, n = mesh_builder.dim, mesh_builder.n_cells
        self.dim = dim
        # gravity (flow_simulation.F90:800-846)
        gin = inp.get("gravity")
        grav = np.zeros(3)
        if isinstance(gin, (list, tuple)):
            grav[: len(gin)] = gin
        else:
            grav[dim - 1] = -(gin if gin is not None else (0.0 if dim == 2 else 9.8))
        # EOS and thermodynamics
        eos = inp.get("eos", "we")
        self.eos = (eos.get("name", "we") if isinstance(eos, dict) else eos).lower()
        temperature = eos.get("temperature", 20.0) if isinstance(eos, dict) else 20.0
        pm = eos.get("permeability_modifier") if isinstance(eos, dict) else None
        self.permeability_modifier = None
        if pm is not None and (pm.get("type", "none").lower() != "none"):
            kind = pm["type"].lower()
            if kind == "power":
                self.permeability_modifier = ("power", [pm.get("exponent", 3.0)])
            elif kind in ("verma-pruess", "verma_pruess"):
                self.permeability_modifier = ("verma-pruess", [pm.get("exponent", 2.0), pm.get("phir", 0.1), pm.get("gamma", 0.7)])
            else:
                raise NotImplementedError("permeability modifier %r" % pm["type"])
        th = inp.get("thermodynamics", "iapws")
        self.thermo = (th.get("name", "iapws") if isinstance(th, dict) else th).lower()
        if self.eos not in ("w", "we", "wce", "wse", "wae", "wsce", "wsae") or self.thermo not in ("iapws", "ifc67"):
            raise NotImplementedError("eos %r / thermodynamics %r" % (self.eos, self.thermo))
        # geometry first without rock (centroids are needed for zones), rock filled in below
        bnds = []
        for b in inp.get("boundaries", []) or []:
            faces = b["faces"]
            for f in (faces if isinstance(faces, list) else [faces]):
                bnds.append((list(f["cells"]), list(f.get("normal", [0.0, 0.0, 1.0])), b["primary"], b.get("region", 1)))
        srcs, self._tables = [], []
        for s in inp.get("source", []) or []:
            bad = [k for k in UNSUPPORTED_SOURCE_KEYS if k in s]
            if bad:
                raise NotImplementedError("source controls %s" % bad)
            if "cell" not in s:
                raise NotImplementedError("sources given by zones or cell lists")
            # rate / enthalpy tables ([[t, value], ...]; setup_table_controls, src/source_setup.F90):
            # averaged over each step interval before the try, see _update_controls
            vals = {}
            for key, default in (("rate", 0.0), ("enthalpy", 83.9e3)):
                v = s.get(key, default)
                if isinstance(v, dict):
                    raise NotImplementedError("source %s given as an object" % key)
                if isinstance(v, (list, tuple)):
                    tab = Table(v, s.get("interpolation", "linear"), s.get("averaging", "integrate"))
                    self._tables.append((len(srcs), key, tab))
                    v = float(tab.interpolate(_get(inp, "time.start", 0.0))[0])
                vals[key] = v
            srcs.append(dict(cell=s["cell"], rate=vals["rate"], enthalpy=vals["enthalpy"],
                             component=s.get("component", 0)))
        minc_in = (mesh or {}).get("minc")
        minc_in = [] if minc_in is None else (minc_in if isinstance(minc_in, list) else [minc_in])
        # matrix cells join their fracture cell's preconditioner subdomain (<= 1024 rows each)
        max_levels = max([len(np.atleast_1d(_get(mz, "geometry.matrix.volume", [0.9]))) for mz in minc_in] or [0])
        if mesh_builder is None:
            lm = unstructured.build_mesh(nodes, cells, dim, thickness=mesh.get("thickness", 1.0),
                                         radial=bool(mesh.get("radial", False)), gravity=grav, boundaries=bnds,
                                         sources=srcs, chunk=512 // (1 + max_levels) if max_levels else 512)
        else:
            lm = mesh_builder(bnds, srcs)
        cen = lm.cell_geom[:n, :3]
        rock = inp.get("rock", {}) or {}
        zones = (mesh or {}).get("zones", {}) or {}
        self._rock_controls = []     # (rock record fields, cells, Table): rock_control.F90:49-116
        for rt in rock.get("types", []) or []:
            rec = rock_record(rt, dim)
            sel = []
            if "cells" in rt and rt["cells"] is not None:
                sel.append(np.asarray(rt["cells"], dtype=int))
            if "zones" in rt and rt["zones"] is not None:
                for z in ([rt["zones"]] if isinstance(rt["zones"], str) else rt["zones"]):
                    sel.append(zone_cells(zones.get(z) if z in zones else (None if z == "all" else zones[z]), cen))
            for idx in sel:
                lm.rock[idx] = rec
            if sel and (_is_table(rt.get("permeability")) or isinstance(rt.get("porosity"), (list, tuple))):
                cells = np.unique(np.concatenate(sel)).astype(np.int32)
                interp = rt.get("interpolation", "linear")
                if _is_table(rt.get("permeability")):
                    tab = Table(rt["permeability"], interpolation=interp)
                    if tab.dim not in (1, dim):
                        raise ValueError("permeability table: 1 or %d values per row" % dim)
                    self._rock_controls.append(((0, 1, 2), cells, tab))
                if isinstance(rt.get("porosity"), (list, tuple)):
                    por = rt["porosity"] if _is_table(rt["porosity"]) else [rt["porosity"]]
                    self._rock_controls.append(((5,), cells, Table(por, interpolation=interp)))
        for k in range(lm.n_bc):
            lm.rock[n + k] = lm.rock[lm.face_cells[lm.n_faces - lm.n_bc + k, 0]]
        # MINC zones (setup of src/minc.F90:58-374): the zone's cells become fracture cells with
        # nested matrix cells behind them
        self._order = None
        if minc_in and self._rock_controls:
            raise NotImplementedError("rock table controls together with MINC zones")
        if minc_in:
            from . import mesh as M
            by_name = {rt.get("name", "").strip(): rt for rt in rock.get("types", []) or []}
            zlist = []
            for mz in minc_in:
                fv = _get(mz, "geometry.fracture.volume")
                mv = list(np.atleast_1d(_get(mz, "geometry.matrix.volume", [0.9])))
                fv = 1.0 - sum(mv) if fv is None else fv
                planes = _get(mz, "geometry.fracture.planes", 1)
                spc = _get(mz, "geometry.fracture.spacing", 50.0)
                spc = [float(spc)] * planes if np.isscalar(spc) else (list(spc) + [spc[0]] * planes)[:planes]
                geo = M.MincGeometry([fv] + mv, spc, _get(mz, "geometry.fracture.connection", 0.0))
                rk = mz.get("rock", {}) or {}
                if isinstance(rk, list):
                    raise NotImplementedError("several rock specifications in one MINC zone")
                sel = []
                if rk.get("zones") is not None:
                    for zn in ([rk["zones"]] if isinstance(rk["zones"], str) else rk["zones"]):
                        sel.append(zone_cells(zones[zn], cen))
                if rk.get("types") is not None:
                    for tn in ([rk["types"]] if isinstance(rk["types"], str) else rk["types"]):
                        rt = by_name[tn.strip()]
                        if rt.get("cells") is not None:
                            sel.append(np.asarray(rt["cells"], dtype=int))
                        for zn in ([rt["zones"]] if isinstance(rt.get("zones"), str) else rt.get("zones") or []):
                            sel.append(zone_cells(zones[zn], cen))
                zc = np.unique(np.concatenate(sel)) if sel else np.arange(n)

                def named(key):
                    tn = _get(rk, key + ".type")
                    return rock_record(by_name[tn.strip()], dim) if tn is not None else None
                mrock = named("matrix")
                if mrock is None:
                    raise NotImplementedError("MINC matrix rock given by properties instead of a rock type")
                zlist.append(dict(cells=zc, geometry=geo, matrix_rock=mrock, fracture_rock=named("fracture")))
            lm = M.add_minc_zones(lm, zlist)
            self._order = lm.extras["waiwera_order"]
        self.owned_gid = np.arange(lm.n_owned)
        self._src_pick = None
        if self.world > 1:
            if minc_in or self._rock_controls or inp.get("tracer") or inp.get("network"):
                raise NotImplementedError("MINC zones, rock table controls, tracers and source networks of an input file on several ranks")
            from .partition import block_owner, partition_mesh
            own = block_owner(lm.n_owned, self.world) if owner is None else np.asarray(owner)
            lm, self._gid = partition_mesh(lm, own, self.rank, world=self.world)
            self.owned_gid = lm.owned_gid
            self._src_pick = lm.extras.get("src_global_index", np.zeros(0, dtype=np.int32))
            self._tables = [(int(np.nonzero(self._src_pick == i)[0][0]), key, tab) for (i, key, tab) in self._tables
                            if i in set(self._src_pick.tolist())]
        self.mesh = lm
        self.relperm = relperm_spec(rock.get("relative_permeability"))
        self.capillary = capillary_spec(rock.get("capillary_pressure"))
        # initial conditions
        init = inp.get("initial", {}) or {}
        npv = {"w": 1, "we": 2, "wce": 3, "wse": 3, "wae": 3, "wsce": 4, "wsae": 4}[self.eos]
        if "filename" in init:
            # restart from a Waiwera HDF5 output (setup_initial, src/initial.F90:421-677, 776, 922):
            # primaries of each cell from its fluid fields by region (eos%primary_variables)
            from . import hdf5io
            st = hdf5io.read_state(os.path.join(base_dir, init["filename"]), init.get("index", -1))
            region = np.rint(st["fluid_region"]).astype(np.int32) if "fluid_region" in st else np.ones(n, dtype=np.int32)
            cols = [st["fluid_pressure"]]
            if npv > 1:
                cols.append(np.where(region == 4, st["fluid_vapour_saturation"], st["fluid_temperature"]))
            if npv > 3:
                raise NotImplementedError("restart files for eos %s" % self.eos)
            if npv > 2 and self.eos == "wse":
                halite = np.isin(region, (5, 6, 8))
                cols[1] = np.where(np.isin(region, (4, 8)), st["fluid_vapour_saturation"], st["fluid_temperature"])
                cols.append(np.where(halite, st["fluid_solid_saturation"], st["fluid_liquid_salt_mass_fraction"]))
            elif npv > 2:
                cols.append(st["fluid_CO2_partial_pressure" if self.eos == "wce" else "fluid_air_partial_pressure"])
            prim = np.stack(cols, axis=1)
            if prim.shape[0] != n and not (self._order is not None and prim.shape[0] == lm.n_owned):
                raise ValueError("initial conditions file has %d cells, mesh has %d" % (prim.shape[0], n))
        else:
            prim = np.asarray(init.get("primary", [1.0e5, 20.0, 0.0][:npv]), dtype=np.float64)
            prim = np.tile(prim, (n, 1)) if prim.ndim == 1 else prim
            region = np.asarray(init.get("region", 1))
            region = np.full(n, int(region), dtype=np.int32) if region.ndim == 0 else region.astype(np.int32)
        if self._order is not None:
            # matrix cells start from their fracture cell's state (src/initial.F90: MINC cells copy
            # the original cell's values unless the file holds them); file order = reference order
            nt = lm.n_owned
            src = np.empty(nt, dtype=np.int64)
            if prim.shape[0] == nt:        # restart file of a MINC run: already one record per cell
                src[self._order] = np.arange(nt)
            else:
                src[:] = lm.extras["minc_parent"]
            prim, region = prim[src], region[src]
        if self.world > 1:
            prim, region = prim[self._gid], region[self._gid]
        self.primary, self.region = prim, region
        # the flow object
        if ode_factory is None:
            from .flow_simulation import FlowSimulation
            self.ode = FlowSimulation(lm, eos=self.eos, device=device, temperature=temperature,
                                      relperm=self.relperm, capillary=self.capillary, thermo=self.thermo,
                                      permeability_modifier=self.permeability_modifier)
        elif self.permeability_modifier is not None:
            self.ode = ode_factory(lm, self.eos, self.thermo, self.relperm, self.capillary, temperature,
                                   permeability_modifier=self.permeability_modifier)
        else:
            self.ode = ode_factory(lm, self.eos, self.thermo, self.relperm, self.capillary, temperature)
        self.ode.set_regions(region)
        if self.world > 1:
            if comm_id is None:
                raise ValueError("several ranks need the communicator id of rank 0 (waiwera_amd.lib.comm_unique_id)")
            self.ode.comm_init(self.rank, self.world, comm_id)
        self.y = np.ascontiguousarray(self.ode.scale(prim, region).ravel())
        # solver and time stepping parameters
        step = _get(inp, "time.step", {}) or {}
        nl = _get(step, "solver.nonlinear", {}) or {}
        opts = {}
        if _get(nl, "tolerance.function.relative") is not None:
            opts["ftol_rel"] = nl["tolerance"]["function"]["relative"]
        if _get(nl, "tolerance.function.absolute") is not None:
            opts["ftol_abs"] = nl["tolerance"]["function"]["absolute"]
        if _get(nl, "maximum.iterations") is not None:
            opts["max_newton_its"] = nl["maximum"]["iterations"]
        if _get(nl, "minimum.iterations") is not None:
            opts["min_newton_its"] = nl["minimum"]["iterations"]
        lin = _get(step, "solver.linear", {}) or {}
        if lin.get("type") in ("bcgs", "gmres", "bcgsl", "lgmres"):
            opts["ksp_type"] = lin["type"]
        elif lin.get("type") is not None:
            raise NotImplementedError("linear solver type %r" % lin["type"])
        if _get(lin, "tolerance.relative") is not None:
            opts["ksp_rtol"] = lin["tolerance"]["relative"]
        if _get(lin, "maximum.iterations") is not None:
            opts["ksp_max_its"] = lin["maximum"]["iterations"]
        if _get(lin, "options.gmres.restart") is not None:
            opts["gmres_restart"] = lin["options"]["gmres"]["restart"]
        # preconditioner (src/timestepper.F90:1745-1757, default "asm"); "ilu" of a serial run is the
        # one-block case of either.  Sub-preconditioner ilu with "factor.levels" k (ILU(k), :1716-1718, 1827) or lu.
        # An input that names no preconditioner gets the REFERENCE's default, restricted PCASM with overlap 1 over ILU(0)
        # (default_flow_pc_type_str = "asm", src/timestepper.F90:2019-2020) -- not the library's own default
        # (wai_default_opts: brick block Jacobi, the fused fast path).  Measured on the 216^3 bench system (round 6,
        # profiles/pc_compare_r6.log): asm 74 Krylov iterations at 7.83 ms, bjacobi 98 at 1.39 ms; a host that wants the
        # fast path for an unmodified input passes default_pc="bjacobi".  pc_choice records what was taken and why.
        pct_in = _get(lin, "preconditioner.type")
        pct = (pct_in or self.default_pc).lower()
        if pct not in ("asm", "bjacobi", "ilu", "lu", "none"):
            raise NotImplementedError("preconditioner type %r" % pct)
        self.pc_choice = (pct, "input" if pct_in else ("reference default" if self.default_pc == "asm" else "default_pc argument"))
        opts["pc_type"] = {"ilu": "bjacobi"}.get(pct, pct)
        sub = _get(lin, "preconditioner.sub.preconditioner", {}) or {}
        subt = (sub.get("type") or "ilu").lower()
        if subt == "lu" and pct in ("bjacobi", "asm"):
            opts["pc_type"] = "lu"       # exact block solves (block Jacobi; no overlap)
        elif subt != "ilu":
            raise NotImplementedError("sub-preconditioner %r" % (sub,))
        else:
            levels = int(_get(sub, "factor.levels") or 0)
            if levels and opts["pc_type"] not in ("asm", "bjacobi"):
                raise NotImplementedError("factor.levels needs a block Jacobi or ASM preconditioner")
            opts["ilu_levels"] = levels
        if opts:
            self.ode.set_opts(**opts)
        # tracers
        tr = inp.get("tracer")
        self.tracer_names = []
        self.X = None
        if tr:
            tr = tr if isinstance(tr, list) else [tr]
            phases = [{"liquid": 0, "vapour": 1}[t.get("phase", "liquid")] for t in tr]
            self.tracer_names = [t.get("name", "tracer") for t in tr]
            nt = len(tr)

            def tvals(v):
                v = 0.0 if v is None else v
                return [v] * nt if np.isscalar(v) else list(v)
            bc = np.array([tvals(b.get("tracer")) for b in inp.get("boundaries", []) or []
                           for f in (b["faces"] if isinstance(b["faces"], list) else [b["faces"]]) for _ in f["cells"]])
            # tracer injection rates: numbers, or [[t, q], ...] tables averaged over each step interval
            # like the rate tables (tracer table source controls, src/source_setup.F90:2415-2560)
            self._tracer_tables, rows = [], []
            for i, s in enumerate(inp.get("source", []) or []):
                v = s.get("tracer")
                if isinstance(v, (list, tuple)) and v and isinstance(v[0], (list, tuple)):
                    if nt != 1:
                        raise NotImplementedError("tracer injection tables with several tracers")
                    tab = Table(v, s.get("interpolation", "linear"), s.get("averaging", "integrate"))
                    self._tracer_tables.append((i, 0, tab))
                    v = float(tab.interpolate(_get(inp, "time.start", 0.0))[0])
                rows.append(tvals(v))
            inj = np.array(rows) if srcs else None
            self._tracer_injection = inj
            self.ode.set_tracers(phases, decay=[t.get("decay", 0.0) for t in tr],
                                 activation=[t.get("activation", 0.0) for t in tr],
                                 diffusion=[t.get("diffusion", 0.0) for t in tr],
                                 bc=bc if lm.n_bc else None, injection=inj)
            self.X = np.tile(np.asarray(tvals(init.get("tracer")), dtype=np.float64), lm.n_owned)
        ad = step.get("adapt", {}) or {}
        mx = step.get("maximum", {}) or {}
        self.ts = Timestepper(
            self.ode, self.y, time=_get(inp, "time.start", 0.0), stepsize=step.get("size", 0.1),
            method=step.get("method", "beuler"), adapt=bool(ad.get("on", False)),
            adapt_method=ad.get("method", "iteration"), adapt_min=ad.get("minimum", 5.0),
            adapt_max=ad.get("maximum", 8.0), reduction=ad.get("reduction", 0.2),
            amplification=ad.get("amplification", 2.0), max_stepsize=mx.get("size") or 0.0,
            max_num_tries=_get(step, "maximum.tries", 10), stop_time=_get(inp, "time.stop"),
            max_num_steps=mx.get("number") if mx.get("number") is not None else 100, aux_solution=self.X,
            checkpoints=_get(inp, "output.checkpoint.time"),
            checkpoint_tolerance=_get(inp, "output.checkpoint.tolerance", 0.1))
        if _get(inp, "output.checkpoint.repeat") not in (None, False, 1):
            raise NotImplementedError("repeated output checkpoints")

        src_in = inp.get("source", []) or []
        # On several ranks the initial fluid state some controls need (reference pressure "initial", productivity index
        # from the initial rate) comes out of a pre_eval, and a pre_eval there exchanges halos: it is made on EVERY rank,
        # decided on the unfiltered source list -- a rank without such a source would otherwise never issue the matching
        # exchange (advisor, round 5)
        need_fluid = self.world > 1 and any(k in s_ for s_ in src_in for k in ("deliverability", "recharge", "injectivity"))
        if self._src_pick is not None:      # the sources of this rank's cells, in the rank's order
            src_in = [src_in[i] for i in self._src_pick]
        self._setup_source_controls(src_in, _get(inp, "time.start", 0.0), collective_fluid=need_fluid)
        self._setup_network(inp)
        if (self._tables or self._ctl_tables or getattr(self, "_tracer_tables", None) or getattr(self, "_network_timed", False)
                or self._rock_controls):
            self.ts.controls = self._update_controls

    # ---- source network --------------------------------------------------------------------------
    def _setup_network(self, inp):
        """"network": {"group": [...], "reinject": [...]} handed to wai_set_source_network; descriptions
        with time tables are sent again for every step interval (_update_controls)"""
        t0 = _get(inp, "time.start", 0.0)
        spec, names, timed = network_spec(inp, (t0, t0), self.ode.separator_enthalpies if hasattr(self.ode, "separator_enthalpies") else None)
        self.network_names = names
        self._network_timed = timed
        if spec is not None:
            self.ode.set_source_network(spec)

    # ---- state-dependent source controls -------------------------------------------------------
    def _setup_source_controls(self, sources, t0, collective_fluid=False):
        """Deliverability, recharge, limiter, separator and direction of each source
        (setup_inline_source_controls, src/source_setup.F90:2340-2412) as the control records the
        device evaluates (include/waiwera_hip.h, wai_source_control); their time tables are kept
        here and averaged over every step interval (_update_controls)"""
        self._ctl, self._ctl_tables = None, []
        fl = None
        if collective_fluid:      # every rank together, whether or not it owns such a source
            assert self.ode.pre_eval(t0, self.y) == 0
            fl = np.asarray(self.ode.fluid())
        if not any(k in s for s in sources for k in ("deliverability", "recharge", "injectivity", "limiter",
                                                     "direction", "factor", "separator")):
            return
        recs = [dict() for _ in sources]

        def timed(v, default, s):     # number | [[t, v], ...] | {"time": [[t, v], ...]} -> Table
            if isinstance(v, dict):
                v = v.get("time")
            if v is None:
                v = default
            data = v if isinstance(v, (list, tuple)) else [[0.0, float(v)]]
            return Table(data, s.get("interpolation", "linear"), s.get("averaging", "integrate"))

        def cell_fluid(cell):
            nonlocal fl
            if fl is None:
                if self.world > 1:
                    raise RuntimeError("initial fluid state asked for on one rank only: the evaluation is collective")
                assert self.ode.pre_eval(t0, self.y) == 0
                fl = np.asarray(self.ode.fluid())
            if self.world > 1:      # the input's cell number -> this rank's (the source is on this rank: its cell is owned)
                cell = int(np.nonzero(self.owned_gid == cell)[0][0])
            return fl[cell]

        nc = {"w": 1, "we": 1, "wce": 2, "wse": 2, "wae": 2, "wsce": 3, "wsae": 3}[self.eos]
        f0, pd = 6 + nc, 7 + nc

        def mobility_sum(f):
            phases = int(round(f[4]))
            return sum(f[f0 + p * pd + 3] * f[f0 + p * pd] / f[f0 + p * pd + 1] for p in range(2) if phases & (1 << p))

        for i, s in enumerate(sources):
            r = recs[i]
            for key, kind, cdef, ckey in (("deliverability", "deliverability", 1.0e-11, "productivity"),
                                          ("recharge", "recharge", 0.0, "coefficient"),
                                          ("injectivity", "recharge", 0.0, "coefficient")):   # same control, :3013
                if key not in s:
                    continue
                spec = s[key] if isinstance(s[key], dict) else {}
                thr = float(spec.get("threshold", -1.0) or -1.0)
                if thr > 0.0:
                    # deliverability that switches on below a threshold pressure (source_control.F90:99-100, 489-503;
                    # source_setup.F90:2855-2911): the source keeps its own rate above it
                    if kind != "deliverability":
                        raise ValueError("threshold belongs to a deliverability control")
                    r["threshold"] = thr
                r["kind"] = kind
                pr = spec.get("pressure", 1.0e5)
                if isinstance(pr, str):
                    if pr.lower() != "initial":
                        raise ValueError("reference pressure %r" % pr)
                    pr = float(cell_fluid(s["cell"])[0])      # set_reference_pressure_initial
                if isinstance(pr, dict) and ("enthalpy" in pr or "pressure" in pr):
                    if kind != "deliverability" or s.get("interpolation", "linear") != "linear":
                        raise NotImplementedError("reference pressure table against enthalpy / pressure")
                    r["table_coord"] = "enthalpy" if "enthalpy" in pr else "pressure"
                    r["table"] = [tuple(q) for q in pr[r["table_coord"]]]
                else:
                    self._ctl_tables.append((i, "pressure", timed(pr, 1.0e5, s)))
                if ckey in spec or kind == "recharge" or "rate" not in s or thr > 0.0:
                    self._ctl_tables.append((i, "coef", timed(spec.get(ckey), cdef, s)))
                    if thr > 0.0:   # threshold_productivity starts as the productivity at the start time (:2906-2909)
                        r["threshold_pi"] = float(self._ctl_tables[-1][2].interpolate(t0)[0])
                else:
                    # productivity index from the initial rate (calculate_PI_from_rate,
                    # src/source_control.F90:407-468) on the initial fluid
                    f = cell_fluid(s["cell"])
                    pref = r["table"][0][1] if "table" in r else self._ctl_tables[-1][2].interpolate(t0)[0]
                    factor = mobility_sum(f) * (f[0] - pref)
                    r["coef"] = abs(s["rate"]) / factor if abs(factor) > 1.0e-9 else cdef
            if "limiter" in s:
                lim = s["limiter"]
                if "limit" not in lim and "type" not in lim:
                    kinds = [k for k in ("total", "water", "steam") if k in lim]
                    if len(kinds) != 1:
                        raise NotImplementedError("limiters on several flow types")
                    lim = dict(lim, type=kinds[0], limit=lim[kinds[0]])    # {"total": 2.0} form
                r["limiter"] = lim.get("type", "total")
                ls = dict(s, **{k: lim[k] for k in ("interpolation", "averaging") if k in lim})
                self._ctl_tables.append((i, "limit", timed(lim.get("limit"), 1.0, ls)))
            sep = s.get("separator")
            psep = None
            if sep is not None and sep is not False:
                psep = sep.get("pressure", 0.55e6) if isinstance(sep, dict) else 0.55e6
            elif "limiter" in s and "separator_pressure" in s["limiter"]:
                psep = s["limiter"]["separator_pressure"]
            stages = list(psep) if isinstance(psep, (list, tuple)) else [psep]
            if len(stages) > 4:
                raise NotImplementedError("separators with more than 4 stages")
            has_sep = psep is not None and stages and all(p is not None and p > 0.0 for p in stages)   # separator_init, separator.F90:182
            if has_sep:     # also what the source network separates the source's flow with
                r["sep_hf"], r["sep_hg"] = self.ode.separator_enthalpies(float(stages[0]))
                r["sep_more"] = [self.ode.separator_enthalpies(float(p)) for p in stages[1:]]
            elif r.get("limiter") in ("water", "steam"):
                r["limiter"] = None      # no separator: separated flows are zero, never over the limit
            if "direction" in s:
                r["direction"] = s["direction"].lower()
            if "factor" in s:      # rate factor, applied after every other control (:2615-2660)
                fac = s["factor"]
                fs = dict(s)
                if isinstance(fac, dict):
                    fs.update({k: fac[k] for k in ("interpolation", "averaging") if k in fac})
                self._ctl_tables.append((i, "factor", timed(fac, 1.0, fs)))
        self._ctl = recs
        self._apply_controls((t0, t0))
        for r in self._ctl:
            r.pop("threshold_pi", None)     # from now on the index the device notes (wai_set_source_controls keeps it)

    def _apply_controls(self, interval):
        for i, key, tab in self._ctl_tables:
            self._ctl[i][key] = float(tab.average(interval)[0])
        self.ode.set_source_controls(self._ctl)

    def _update_controls(self, interval):
        """table_object_control_update (src/control.F90:263-284): each table's average over the step
        interval becomes the rate / enthalpy of its source; likewise the productivity, reference
        pressure and limit tables of the state-dependent controls"""
        # rock controls: flow_simulation_pre_try_timestep(t) with t the time the try ends at (timestepper.F90:2333),
        # the table's value AT that time (rock_control.F90:66, 102: interpolate, not an interval average); a scalar
        # permeability fills all directions
        for fields, cells, tab in self._rock_controls:
            v = tab.interpolate(float(interval[1]))
            for q, f in enumerate(fields):
                self.ode.update_rock(f, cells, v[0] if tab.dim == 1 else v[min(q, tab.dim - 1)])
        if self._tables:
            rate, enth = self.mesh.src_rate.copy(), self.mesh.src_enthalpy.copy()
            for i, key, tab in self._tables:
                (rate if key == "rate" else enth)[i] = tab.average(interval)[0]
            self.ode.set_source_rates(rate, enth)
        if self._ctl_tables:
            self._apply_controls(interval)
        if getattr(self, "_network_timed", False):
            spec, _, _ = network_spec(self.inp, interval, self.ode.separator_enthalpies)
            self.ode.set_source_network(spec)
        if getattr(self, "_tracer_tables", None):
            for i, it, tab in self._tracer_tables:
                self._tracer_injection[i, it] = tab.average(interval)[0]
            self.ode.set_tracer_injection(self._tracer_injection)

    @classmethod
    def from_json(cls, path, **kw):
        with open(path) as f:
            inp = json.load(f)
        return cls(inp, base_dir=os.path.dirname(os.path.abspath(path)), **kw)

    def run(self):
        """timestepper_run with the output schedule of "output" (initial / frequency / final,
        src/timestepper.F90:2478-2560); returns the final cell fields under the reference's names
        and, if "output.filename" is given (and the HDF5 library is there), writes that file"""
        assert self.ode.pre_eval(self.ts.time, self.y) == 0
        if self.X is not None:
            self.ts.init_auxiliary()
        oc = self.inp.get("output")
        oc = {} if oc in (None, True) else ({"frequency": 0, "initial": False, "final": False} if oc is False else oc)
        freq, self.outputs = oc.get("frequency", 1), []
        if oc.get("initial", True):
            self.outputs.append(self.fields())
        try:
            while not self.ts.finished:
                self.ts.step()
                hit = self.ts.checkpoint_hit
                if hit or (freq and self.ts.taken % freq == 0) or (self.ts.finished and oc.get("final", True)):
                    self.outputs.append(self.fields())
                if hit:
                    self.ts.checkpoint_update()
        finally:
            # the reference keeps what it has written when a step aborts; a file that cannot be written
            # is reported, not swallowed (the results are still returned)
            if oc.get("filename") and self.outputs and self.world == 1:     # (several ranks: no output file yet; fields())
                try:
                    self.save_hdf5(os.path.join(self.output_dir, oc["filename"]))
                except Exception as e:
                    self.output_error = e
                    import sys
                    print("waiwera_amd: output file %r not written: %s" % (oc["filename"], e), file=sys.stderr)
        return self.fields()

    def save_hdf5(self, path):
        """the collected outputs in the reference's layout: /time, /cell_index, /cell_fields/*,
        /source_fields/source_rate and source_enthalpy, /minc/level and /minc/parent of a MINC mesh"""
        from . import hdf5io
        outs = getattr(self, "outputs", None) or [self.fields()]
        n = self.mesh.n_owned
        data = {"/time": np.array([[o["time"]] for o in outs]), "/cell_index": np.arange(n, dtype=np.int32)[:, None]}
        if self._order is not None:
            # flow_simulation_output_minc_data (src/flow_simulation.F90:2625-2691): per cell, in the output's cell
            # order, its MINC level (0: fracture or single-porosity cell) and the natural index of the original cell
            ex = self.mesh.extras
            data["/minc/level"] = np.asarray(ex["minc_level"])[self._order].astype(np.int32)[:, None]
            data["/minc/parent"] = np.asarray(ex["minc_parent"])[self._order].astype(np.int32)[:, None]
        for k in outs[0]:
            if k == "time":
                continue
            if k.startswith("source_") or k.startswith("network_"):
                data["/source_fields/" + k] = np.stack([o[k] for o in outs])
            elif k.startswith("flux_"):
                data["/face_fields/" + k] = np.stack([o[k] for o in outs])
            elif k.startswith("face_geometry"):
                data["/face_fields/" + k] = outs[0][k]
            elif k.startswith("face_cell_"):
                data["/" + k] = np.asarray(outs[0][k], dtype=np.int32)[:, None]
            elif k.startswith("cell_geometry"):
                data["/cell_fields/" + k] = outs[0][k]
            else:
                data["/cell_fields/" + k] = np.stack([o[k] for o in outs])
        hdf5io.write_file(path, data)

    def fields(self):
        n = self.mesh.n_owned
        self.ode.pre_eval(self.ts.time, self.y)
        fl = np.asarray(self.ode.fluid())[:n]
        geom = self.mesh.cell_geom[:n]
        if self._order is not None:      # MINC: the reference's cell order (original cells, then level by level)
            fl, geom = fl[self._order], geom[self._order]
        nc = {"w": 1, "we": 1, "wce": 2, "wse": 2, "wae": 2, "wsce": 3, "wsae": 3}[self.eos]
        f0, pd = 6 + nc, 7 + nc
        out = {"time": self.ts.time, "fluid_pressure": fl[:, 0].copy(), "fluid_temperature": fl[:, 1].copy(),
               "fluid_region": fl[:, 2].copy(), "fluid_liquid_saturation": fl[:, f0 + 2].copy(),
               "fluid_liquid_density": fl[:, f0].copy(),
               "cell_geometry_centroid": geom[:, : self.dim].copy(),
               "cell_geometry_volume": geom[:, 3].copy()}
        if self.eos != "w":
            out["fluid_vapour_saturation"] = fl[:, f0 + pd + 2].copy()
            out["fluid_vapour_density"] = fl[:, f0 + pd].copy()
        if self.eos in ("wse", "wsce", "wsae"):
            out["fluid_liquid_salt_mass_fraction"] = fl[:, f0 + 8].copy()
            out["fluid_solid_saturation"] = fl[:, f0 + 2 * pd + 2].copy()
        if self.eos in ("wsce", "wsae"):
            gas = "CO2" if self.eos == "wsce" else "air"
            out["fluid_%s_partial_pressure" % gas] = fl[:, 8].copy()
            out["fluid_liquid_%s_mass_fraction" % gas] = fl[:, f0 + 9].copy()
            out["fluid_vapour_%s_mass_fraction" % gas] = fl[:, f0 + pd + 9].copy()
        if self.eos in ("wce", "wae"):
            gas = "CO2" if self.eos == "wce" else "air"
            out["fluid_%s_partial_pressure" % gas] = fl[:, 7].copy()
            out["fluid_liquid_%s_mass_fraction" % gas] = fl[:, f0 + 8].copy()
            out["fluid_vapour_%s_mass_fraction" % gas] = fl[:, f0 + pd + 8].copy()
        if self.X is not None:
            for k, name in enumerate(self.tracer_names):
                xk = self.X.reshape(n, -1)[:, k]
                out["tracer_" + name] = (xk[self._order] if self._order is not None else xk).copy()
        if self.mesh.n_src and hasattr(self.ode, "source_rates"):
            out["source_rate"], out["source_enthalpy"] = self.ode.source_rates()
        oc = self.inp.get("output")
        want = (oc.get("fields") if isinstance(oc, dict) else None) or {}
        # face fields: "output.fields.flux" (src/flow_simulation.F90:460-504): the flux vector's
        # components and phases by name, per unit area, positive from face_cell_1 to face_cell_2
        flux_names = want.get("flux") or []
        if flux_names and hasattr(self.ode, "fluxes"):
            comps = {"w": ["water"], "we": ["water", "energy"], "wce": ["water", "CO2", "energy"],
                     "wae": ["water", "air", "energy"], "wse": ["water", "salt", "energy"],
                     "wsce": ["water", "salt", "CO2", "energy"], "wsae": ["water", "salt", "air", "energy"]}[self.eos]
            names = comps + (["liquid"] if self.eos == "w" else ["liquid", "vapour"])
            fx = self.ode.fluxes()
            for nm in (names if flux_names == "all" or "all" in flux_names else flux_names):
                out["flux_" + nm] = fx[:, names.index(nm)].copy()
            fc = np.asarray(self.mesh.face_cells)
            c2 = fc[:, 1].astype(np.int64)
            bnd = c2 >= self.mesh.n_owned + self.mesh.n_halo
            spec = np.asarray(self.mesh.extras.get("bc_spec", np.zeros(self.mesh.n_bc, dtype=np.int64)))
            c2 = np.where(bnd, -1 - spec[np.clip(c2 - self.mesh.n_owned - self.mesh.n_halo, 0, max(self.mesh.n_bc - 1, 0))], c2)
            out["face_cell_1"], out["face_cell_2"] = fc[:, 0].copy(), c2
            out["face_geometry_area"] = np.asarray(self.mesh.face_geom)[:, 0].copy()
        # separated water / steam flows of the sources and the source network's nodes
        src_want = want.get("source") or []
        if self.mesh.n_src and hasattr(self.ode, "source_separated") and \
                any(k in src_want for k in ("water_rate", "water_enthalpy", "steam_rate", "steam_enthalpy")):
            sep = self.ode.source_separated()
            for j, k in enumerate(("water_rate", "water_enthalpy", "steam_rate", "steam_enthalpy")):
                if k in src_want:
                    out["source_" + k] = sep[:, j].copy()
        if getattr(self, "network_names", None) and hasattr(self.ode, "source_network") and \
                (self.network_names["group"] or self.network_names["reinject"]):
            G, R = self.ode.source_network()
            gcols = ("rate", "enthalpy", "water_rate", "water_enthalpy", "steam_rate", "steam_enthalpy")
            for k in (want.get("network_group") or []):
                if k in gcols and len(G):
                    out["network_group_" + k] = G[:, gcols.index(k)].copy()
            rcols = ("output_water_rate", "output_steam_rate", "overflow_rate", "overflow_enthalpy", "overflow_water_rate",
                     "overflow_water_enthalpy", "overflow_steam_rate", "overflow_steam_enthalpy")
            for k in (want.get("network_reinject") or ["output_water_rate", "output_steam_rate", "overflow_water_rate",
                                                       "overflow_steam_rate"]):
                if k in rcols and len(R):
                    out["network_reinject_" + k] = R[:, rcols.index(k)].copy()
        return out

    def save(self, path):
        np.savez(path, owned_gid=self.owned_gid, **self.fields())
