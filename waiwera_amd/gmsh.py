"""Reader for gmsh MSH 2.2 files (ASCII or binary), the mesh format of the reference's benchmark
inputs ("mesh": {"filename": "*.msh"}, read there through DMPlexCreateFromFile, src/mesh.F90:150).
Only what a finite-volume flow mesh needs: node coordinates and the elements of the highest
dimension present (quadrangles / triangles in 2-D, hexahedra / prisms / tetrahedra in 3-D),
0-based, in file order -- the reference's natural cell order."""
import struct

import numpy as np

# gmsh element type -> (number of nodes, topological dimension)
ELEMENT = {1: (2, 1), 2: (3, 2), 3: (4, 2), 4: (4, 3), 5: (8, 3), 6: (6, 3), 15: (1, 0)}


def read_msh(path):
    """(nodes (N, 3), cells: list of node-index lists, dim)"""
    b = open(path, "rb").read()
    i = b.index(b"$MeshFormat") + len(b"$MeshFormat")
    j = b.index(b"\n", i + 1)
    version, ftype, dsize = b[i:j].split()
    if not version.startswith(b"2"):
        raise ValueError("only MSH 2.x files are supported, got " + version.decode())
    binary = int(ftype) == 1
    p = b.index(b"$Nodes") + len(b"$Nodes")
    p = b.index(b"\n", p) + 1
    q = b.index(b"\n", p)
    n = int(b[p:q])
    p = q + 1
    ids = np.zeros(n, dtype=np.int64)
    xyz = np.zeros((n, 3))
    if binary:
        for k in range(n):
            ids[k], xyz[k, 0], xyz[k, 1], xyz[k, 2] = struct.unpack("<iddd", b[p:p + 28])
            p += 28
    else:
        for k in range(n):
            q = b.index(b"\n", p)
            t = b[p:q].split()
            ids[k], xyz[k] = int(t[0]), [float(v) for v in t[1:4]]
            p = q + 1
    index = {int(g): k for k, g in enumerate(ids)}
    p = b.index(b"$Elements") + len(b"$Elements")
    p = b.index(b"\n", p) + 1
    q = b.index(b"\n", p)
    ne = int(b[p:q])
    p = q + 1
    elems = []
    if binary:
        cnt = 0
        while cnt < ne:
            et, num, nt = struct.unpack("<iii", b[p:p + 12])
            p += 12
            nn = ELEMENT[et][0]
            for _ in range(num):
                vals = struct.unpack("<%di" % (1 + nt + nn), b[p:p + 4 * (1 + nt + nn)])
                p += 4 * (1 + nt + nn)
                elems.append((et, [index[v] for v in vals[1 + nt:]]))
                cnt += 1
    else:
        for _ in range(ne):
            q = b.index(b"\n", p)
            t = [int(v) for v in b[p:q].split()]
            et, nt = t[1], t[2]
            elems.append((et, [index[v] for v in t[3 + nt:]]))
            p = q + 1
    dim = max(ELEMENT[et][1] for et, _ in elems)
    cells = [nodes for et, nodes in elems if ELEMENT[et][1] == dim]
    return xyz, cells, dim
