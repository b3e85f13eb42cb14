"""Reader for MULgraph geometry files (the `g*.dat` files next to the reference's benchmark inputs).

Several of the reference's benchmarks ship their mesh only as an ExodusII file (no reader for that
binary format here) plus the MULgraph geometry it was generated from: columns (polygons in plan)
times layers.  This module turns that description into the nodes / cells the generic finite-volume
geometry (waiwera_amd/unstructured.py) takes, so that those inputs run as they are.  Cell order is
the order the ExodusII file and AUTOUGH2 share (test/benchmark/*/test_*.py compare them cell by
cell without a permutation): layers from the top, inside a layer the columns in file order.

Format (fixed-width records, the public MULgraph / PyTOUGH geometry layout):
  header    type(5) convention(1) atmosphere_type(1) ...
  VERTICES  name(3) x(10) y(10)
  GRID      column name(3) centre_specified(1) num_nodes(2) [xc(10) yc(10)], then one vertex name(3) per line
  CONNECTIONS  column(3) column(3)         (derivable from shared edges: not needed)
  LAYERS    name(3) bottom(10) centre(10); the first record is the atmosphere / top elevation
  SURFACE   column(3) elevation(10)        (varying tops: not supported)
"""
import numpy as np


def read_geometry(path):
    """returns (nodes (N, 3), cells [node index lists, gmsh hexahedron / prism order], 3)"""
    with open(path) as f:
        lines = [ln.rstrip("\n") for ln in f]
    sections, cur = {}, None
    for ln in lines[1:]:
        key = ln.strip().upper()
        if key in ("VERTICES", "GRID", "CONNECTIONS", "LAYERS", "SURFACE", "WELLS"):
            cur = key
            sections[cur] = []
        elif cur is not None and ln.strip():
            sections[cur].append(ln)
    if sections.get("SURFACE"):
        raise NotImplementedError("MULgraph geometry with a SURFACE section")
    verts = {}
    for ln in sections["VERTICES"]:
        verts[ln[0:3].strip()] = (float(ln[3:13]), float(ln[13:23]))
    columns, k = [], 0
    grid = sections["GRID"]
    while k < len(grid):
        nn = int(grid[k][4:6])
        columns.append([grid[k + 1 + q][0:3].strip() for q in range(nn)])
        k += 1 + nn
    layers = [(float(ln[3:13]), float(ln[13:23])) for ln in sections["LAYERS"]]
    tops = [layers[0][0]] + [b for b, _ in layers[1:-1]]
    bottoms = [b for b, _ in layers[1:]]
    vnames = list(verts)
    vindex = {nm: i for i, nm in enumerate(vnames)}
    nv = len(vnames)
    elevations = [layers[0][0]] + bottoms            # node levels, top first
    nodes = np.array([[verts[nm][0], verts[nm][1], z] for z in elevations for nm in vnames])
    cells = []
    for lay in range(len(bottoms)):
        for col in columns:
            if len(col) not in (3, 4):
                raise NotImplementedError("columns with %d nodes" % len(col))
            xy = np.array([verts[nm] for nm in col])
            x, y = xy[:, 0], xy[:, 1]
            signed = 0.5 * np.sum(x * np.roll(y, -1) - np.roll(x, -1) * y)
            ring = col if signed > 0.0 else col[::-1]      # counter-clockwise seen from above
            lower = [(lay + 1) * nv + vindex[nm] for nm in ring]
            upper = [lay * nv + vindex[nm] for nm in ring]
            cells.append(lower + upper)
    assert len(tops) == len(bottoms)
    return nodes, cells, 3
