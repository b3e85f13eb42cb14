"""Host logic: structured mesh generator, brick numbering, partition and halo lists."""
import numpy as np
import pytest

from waiwera_amd import mesh as M


def build_all(dims, part, brick, order="hyperplane"):
    g = M.StructuredGrid(dims, part=part, brick=brick, order=order)
    return g, [g.local_mesh(r, top_bc=([1e5, 20.0], 1), sources=M.benchmark_sources(g)) for r in range(g.nranks)]


@pytest.mark.parametrize("dims,part,brick", [((8, 8, 8), (1, 1, 1), (4, 4, 4)), ((10, 9, 7), (2, 1, 1), (4, 4, 4)),
                                             ((12, 8, 8), (2, 2, 1), (4, 4, 4)), ((8, 8, 8), (2, 2, 2), (4, 4, 4))])
def test_partition_covers_mesh_and_halos_match(dims, part, brick):
    g, ms = build_all(dims, part, brick)
    gids = np.concatenate([m.owned_gid for m in ms])
    assert np.array_equal(np.sort(gids), np.arange(g.n_global))
    nfaces_interior = sum(int(np.sum((m.face_cells[:, 0] < m.n_owned) & (m.face_cells[:, 1] < m.n_owned)))
                          for m in ms)
    nfaces_cut = sum(int(np.sum((m.face_cells.max(axis=1) >= m.n_owned) & (m.face_cells.max(axis=1) < m.n_prim)))
                     for m in ms)
    nx, ny, nz = dims
    total = (nx - 1) * ny * nz + nx * (ny - 1) * nz + nx * ny * (nz - 1)
    assert nfaces_interior + nfaces_cut // 2 == total
    for r, m in enumerate(ms):
        pg = m.extras["prim_gid"]
        for q, nb in enumerate(m.nbr_ranks):
            mine = pg[m.n_owned + m.recv_ptr[q]: m.n_owned + m.recv_ptr[q + 1]]
            o = ms[nb]
            qq = list(o.nbr_ranks).index(r)
            theirs = o.owned_gid[o.send_idx[o.send_ptr[qq]: o.send_ptr[qq + 1]]]
            assert np.array_equal(mine, theirs)
        # every bc face: ghost cell second, zero volume, d2 = 0, d12 = d1
        if m.n_bc:
            bf = m.face_cells[:, 1] >= m.n_prim
            assert bf.sum() == m.n_bc
            assert np.all(m.cell_geom[m.n_prim:, 3] == 0.0)
            assert np.all(m.face_geom[bf, 2] == 0.0) and np.all(m.face_geom[bf, 3] == m.face_geom[bf, 1])
            assert np.all(m.face_geom[bf, 7] == -M.GRAVITY)


def test_bricks_are_contiguous_and_level_sorted():
    g, (m,) = build_all((10, 9, 7), (1, 1, 1), (4, 4, 4))
    sp = m.sub_ptr
    assert sp[0] == 0 and sp[-1] == m.n_owned
    for s in range(len(sp) - 1):
        c = m.owned_ijk[sp[s]: sp[s + 1]].astype(int)
        ext = c.max(axis=0) - c.min(axis=0) + 1
        assert np.prod(ext) == len(c) and np.all(ext <= 4)
        lv = (c - c.min(axis=0)).sum(axis=1)
        assert np.all(np.diff(lv) >= 0)


def test_face_geometry_conventions():
    g, (m,) = build_all((4, 3, 5), (1, 1, 1), (4, 4, 4), order="natural")
    fg, fc = m.face_geom, m.face_cells
    cen = m.cell_geom[:, :3]
    interior = fc[:, 1] < m.n_prim
    d = cen[fc[interior, 1]] - cen[fc[interior, 0]]
    n = fg[interior, 4:7]
    # normal points from cell 1 to cell 2, distance12 = |d|, g.n = (0,0,-9.8).n
    assert np.allclose((d * n).sum(axis=1), fg[interior, 3])
    assert np.allclose(fg[interior, 7], -M.GRAVITY * n[:, 2])
    assert np.allclose(fg[interior, 1] + fg[interior, 2], fg[interior, 3])
    assert set(np.unique(fg[:, 11])) <= {1.0, 2.0, 3.0}


def test_partition_shape():
    assert M.partition_shape(8) == (2, 2, 2) and M.partition_shape(4) == (2, 2, 1)
    assert M.partition_shape(2) == (2, 1, 1) and M.partition_shape(1) == (1, 1, 1)
    assert np.prod(M.partition_shape(6)) == 6
