"""Raw binary exchange with the Fortran driver (waiwera_amd/fortran/newton_driver.F90)."""
import numpy as np

from .lib import EOS_KIND


def write_case(path, mesh, eos, y, region):
    """Header of 10 int32, then the flat arrays in the order the driver reads them."""
    np_ = {"w": 1, "we": 2, "wce": 3}[eos]
    nsub = mesh.sub_ptr.size - 1
    # hdr[8], hdr[9]: neighbour ranks and cells sent per halo exchange (0: one rank) -- the ghost lists follow the sources
    n_nbr = int(mesh.nbr_ranks.size) if (mesh.n_halo and mesh.nbr_ranks is not None) else 0
    n_send = int(mesh.send_idx.size) if n_nbr else 0
    hdr = np.array([EOS_KIND[eos], mesh.n_owned, mesh.n_halo, mesh.n_bc, mesh.n_faces, nsub,
                    mesh.n_src, np_, n_nbr, n_send], dtype=np.int32)
    with open(path, "wb") as f:
        hdr.tofile(f)
        np.ascontiguousarray(mesh.face_cells, dtype=np.int32).tofile(f)
        np.ascontiguousarray(mesh.face_geom, dtype=np.float64).tofile(f)
        np.ascontiguousarray(mesh.cell_geom, dtype=np.float64).tofile(f)
        np.ascontiguousarray(mesh.rock, dtype=np.float64).tofile(f)
        np.ascontiguousarray(mesh.sub_ptr, dtype=np.int32).tofile(f)
        np.ascontiguousarray(region, dtype=np.int32).tofile(f)
        np.ascontiguousarray(y, dtype=np.float64)[: mesh.n_owned * np_].tofile(f)
        if mesh.n_bc:
            np.ascontiguousarray(mesh.bc_primary, dtype=np.float64).tofile(f)
            np.ascontiguousarray(mesh.bc_region, dtype=np.int32).tofile(f)
        if mesh.n_src:
            np.ascontiguousarray(mesh.src_cell, dtype=np.int32).tofile(f)
            np.ascontiguousarray(mesh.src_rate, dtype=np.float64).tofile(f)
            np.ascontiguousarray(mesh.src_enthalpy, dtype=np.float64).tofile(f)
            np.ascontiguousarray(mesh.src_component, dtype=np.int32).tofile(f)
        if n_nbr:   # wai_set_halo's arguments
            np.ascontiguousarray(mesh.nbr_ranks, dtype=np.int32).tofile(f)
            np.ascontiguousarray(mesh.send_ptr, dtype=np.int32).tofile(f)
            np.ascontiguousarray(mesh.send_idx, dtype=np.int32).tofile(f)
            np.ascontiguousarray(mesh.recv_ptr, dtype=np.int32).tofile(f)


def read_result(path, n_owned, n_prim, np_):
    with open(path, "rb") as f:
        tot = np.fromfile(f, dtype=np.int32, count=2)
        y = np.fromfile(f, dtype=np.float64, count=n_owned * np_)
        region = np.fromfile(f, dtype=np.int32, count=n_prim)
    return int(tot[0]), int(tot[1]), y, region
