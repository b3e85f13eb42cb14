"""Finite-volume geometry of an unstructured mesh (2-D polygons or 3-D polyhedra given by their
nodes) in the form the hot path takes: flat face / cell / rock arrays with the reference's
conventions.

What is restated here is what the reference obtains from DMPlexComputeGeometryFVM and then edits
(src/mesh.F90:462-579): cell centroid and volume, face centroid / area / unit normal pointing from
the face's first to its second cell, 2-D meshes extruded by `thickness` or revolved (`radial`,
Pappus: :341-432), normal distances from the two centroids to the face scaled so that they add up
to the centroid-to-centroid normal distance (non-orthogonal correction, src/face.F90:230-249), the
gravity term g.n, the permeability direction = coordinate axis closest to the normal (:210-226).
Open boundary faces (src/mesh.F90:583-664, 1772-1800): for a listed cell, its boundary face whose
outward normal is closest to the given vector; Dirichlet ghost cell with distances (d, 0)."""
import numpy as np

from .mesh import LocalMesh, default_rock

# node numbering of the faces of gmsh hexahedra / prisms / tetrahedra (outward or inward: the
# orientation is fixed from the centroids below)
HEX_FACES = [(0, 3, 2, 1), (4, 5, 6, 7), (0, 1, 5, 4), (1, 2, 6, 5), (2, 3, 7, 6), (3, 0, 4, 7)]
PRISM_FACES = [(0, 2, 1), (3, 4, 5), (0, 1, 4, 3), (1, 2, 5, 4), (2, 0, 3, 5)]
TET_FACES = [(0, 2, 1), (0, 1, 3), (1, 2, 3), (2, 0, 3)]


def _polygon(xy):
    """area and centroid of a planar polygon given by its vertices in order (shoelace)"""
    x, y = xy[:, 0], xy[:, 1]
    x1, y1 = np.roll(x, -1), np.roll(y, -1)
    cr = x * y1 - x1 * y
    a = 0.5 * cr.sum()
    cx = ((x + x1) * cr).sum() / (6.0 * a)
    cy = ((y + y1) * cr).sum() / (6.0 * a)
    return abs(a), np.array([cx, cy])


def _face3d(p):
    """area vector and centroid of a (nearly) planar 3-D polygon: fan about the vertex average"""
    c0 = p.mean(axis=0)
    av, cen, tot = np.zeros(3), np.zeros(3), 0.0
    for k in range(len(p)):
        a, b = p[k], p[(k + 1) % len(p)]
        v = 0.5 * np.cross(a - c0, b - c0)
        w = np.linalg.norm(v)
        av += v
        cen += w * (a + b + c0) / 3.0
        tot += w
    return av, cen / tot


def build_mesh(nodes, cells, dim, thickness=1.0, radial=False, gravity=None, boundaries=(), rock=None,
               sources=None, chunk=512):
    """boundaries: [(cells, normal, primary, region)]; rock: (8,) or (ncells, 8); sources:
    [{cell, rate, enthalpy, component}].  Returns a LocalMesh (single rank, natural cell order,
    preconditioner subdomains = chunks of consecutive cells)."""
    n = len(cells)
    g = np.zeros(3)
    if gravity is not None:
        g[: len(gravity)] = gravity
    cen = np.zeros((n, 3))
    vol = np.zeros(n)
    facemap = {}        # sorted node tuple -> [cell, ...] and geometry
    fgeo = {}
    for c, nd in enumerate(cells):
        p = nodes[nd]
        if dim == 2:
            a, cc = _polygon(p[:, :2])
            cen[c, :2] = cc
            vol[c] = a
            edges = [(nd[k], nd[(k + 1) % len(nd)]) for k in range(len(nd))]
            for e in edges:
                key = tuple(sorted(e))
                facemap.setdefault(key, []).append(c)
                if key not in fgeo:
                    a0, a1 = nodes[e[0], :2], nodes[e[1], :2]
                    t = a1 - a0
                    length = np.linalg.norm(t)
                    nrm = np.array([t[1], -t[0], 0.0]) / length
                    fgeo[key] = [length, np.array([*(0.5 * (a0 + a1)), 0.0]), nrm]
        else:
            table = {8: HEX_FACES, 6: PRISM_FACES, 4: TET_FACES}[len(nd)]
            fl = []
            for f in table:
                key = tuple(sorted(nd[k] for k in f))
                facemap.setdefault(key, []).append(c)
                if key not in fgeo:
                    av, fc = _face3d(nodes[[nd[k] for k in f]])
                    area = np.linalg.norm(av)
                    fgeo[key] = [area, fc, av / area]
                fl.append(key)
            # volume / centroid: pyramids from the vertex average to each face
            c0 = p.mean(axis=0)
            v, cc = 0.0, np.zeros(3)
            for key in fl:
                area, fc, nrm = fgeo[key]
                h = abs(np.dot(fc - c0, nrm))
                pv = area * h / 3.0
                v += pv
                cc += pv * (0.75 * fc + 0.25 * c0)
            vol[c] = v
            cen[c] = cc / v
    if dim == 2:   # modify_cell_geometry (:341-385)
        vol = vol * (2.0 * np.pi * cen[:, 0] if radial else thickness)

    def face_record(key, c1, c2, x2=None):
        area, fc, nrm = fgeo[key]
        nrm = nrm.copy()
        ref = (cen[c2] if c2 is not None else fc) - cen[c1]
        if np.dot(ref, nrm) < 0:
            nrm = -nrm
        if dim == 2:   # modify_face_geometry (:390-432)
            area = area * (2.0 * np.pi * fc[0] if radial else thickness)
        rec = np.zeros(12)
        rec[0] = area
        if c2 is not None:
            d1, d2 = np.dot(fc - cen[c1], nrm), np.dot(cen[c2] - fc, nrm)
            d12 = np.dot(cen[c2] - cen[c1], nrm)
            corr = d12 / (d1 + d2)
            rec[1], rec[2], rec[3] = d1 * corr, d2 * corr, d12
        else:
            d1 = np.dot(fc - cen[c1], nrm)
            rec[1], rec[2], rec[3] = d1, 0.0, d1
        rec[4:7] = nrm
        rec[7] = np.dot(g, nrm)
        rec[8:11] = fc
        rec[11] = int(np.argmax(np.abs(nrm[:dim]))) + 1
        return rec

    fc_list, fg_list = [], []
    for key, cs in facemap.items():
        if len(cs) == 2:
            fc_list.append((cs[0], cs[1]))
            fg_list.append(face_record(key, cs[0], cs[1]))
    # open boundaries
    bc_prim, bc_region, bc_cen = [], [], []
    cell_bfaces = {}
    for key, cs in facemap.items():
        if len(cs) == 1:
            cell_bfaces.setdefault(cs[0], []).append(key)
    for (bcells, normal, primary, region) in boundaries:
        nv = np.zeros(3)
        nv[: len(normal)] = normal
        for c in bcells:
            best, bkey = -2.0, None
            for key in cell_bfaces.get(c, []):
                area, fc, nrm = fgeo[key]
                out = nrm if np.dot(fc - cen[c], nrm) > 0 else -nrm
                cosv = np.dot(nv[:dim], out[:dim]) / (np.linalg.norm(nv[:dim]) * np.linalg.norm(out[:dim]))
                if cosv > best:
                    best, bkey = cosv, key
            if bkey is None:
                raise ValueError("cell %d has no boundary face" % c)
            ghost = n + len(bc_prim)
            fc_list.append((c, ghost))
            fg_list.append(face_record(bkey, c, None))
            bc_prim.append(np.asarray(primary, dtype=np.float64))
            bc_region.append(int(region))
            bc_cen.append(fgeo[bkey][1])
    m = LocalMesh(dims=(n, 1, 1), spacing=(0.0, 0.0, 0.0), part=(1, 1, 1), rank=0, brick=(chunk, 1, 1), n_global=n)
    m.n_owned, m.n_halo, m.n_bc = n, 0, len(bc_prim)
    m.face_cells = np.asarray(fc_list, dtype=np.int32).reshape(-1, 2)
    m.face_geom = np.asarray(fg_list).reshape(-1, 12)
    m.n_faces = m.face_cells.shape[0]
    cg = np.zeros((n + m.n_bc, 4))
    cg[:n, :3], cg[:n, 3] = cen, vol
    for k, fcen in enumerate(bc_cen):
        cg[n + k, :3] = fcen
    m.cell_geom = cg
    rk = np.zeros((n + m.n_bc, 8))
    rk[:n] = default_rock(1)[0] if rock is None else np.asarray(rock, dtype=np.float64)
    for k in range(m.n_bc):
        rk[n + k] = rk[m.face_cells[m.n_faces - m.n_bc + k, 0]]
    m.rock = rk
    if m.n_bc:
        m.bc_primary = np.asarray(bc_prim)
        m.bc_region = np.asarray(bc_region, dtype=np.int32)
    m.sub_ptr = np.append(np.arange(0, n, chunk), n).astype(np.int32)
    m.owned_gid = np.arange(n)
    m.nbr_ranks = np.zeros(0, dtype=np.int32)
    m.send_ptr = np.zeros(1, dtype=np.int32)
    m.send_idx = np.zeros(0, dtype=np.int32)
    m.recv_ptr = np.zeros(1, dtype=np.int32)
    if sources:
        m.n_src = len(sources)
        m.src_cell = np.array([s["cell"] for s in sources], dtype=np.int32)
        m.src_rate = np.array([s["rate"] for s in sources], dtype=np.float64)
        m.src_enthalpy = np.array([s.get("enthalpy", 0.0) for s in sources], dtype=np.float64)
        m.src_component = np.array([s.get("component", 0) for s in sources], dtype=np.int32)
    return m
