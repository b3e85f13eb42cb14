"""Partition of ANY one-rank mesh (LocalMesh: structured, gmsh, MULgraph, MINC) into the flat arrays of one rank of an
N-rank run -- what DMPlexDistribute with an overlap of one cell gives the reference (src/mesh.F90:143-171): every
rank owns a set of cells, carries one layer of the neighbouring ranks' cells as ghosts and computes the fluxes through
the faces it shares with them itself (src/flow_simulation.F90:1449-1450); the ghosts' values arrive through the halo
lists (wai_set_halo; DMGlobalToLocal, src/dm_utils.F90:480-498).

The structured builder (mesh.StructuredGrid.local_mesh) produces a rank's arrays directly and never holds the global
mesh; this module is for the inputs that are read whole (tests/golden/inputs/*.msh, *.dat: a few hundred cells) and
for checking the structured builder against an independent construction.

Conventions (the library's, include/waiwera_hip.h): local cell order [owned | ghosts grouped by owning rank | Dirichlet
boundary cells]; the ghosts of neighbour q are q's cells next to mine in ascending global index, which is also the order
in which q lists them in its send list for me; faces keep the one-rank mesh's order and orientation."""
import copy

import numpy as np


def block_owner(n_cells, world):
    """contiguous blocks of the global numbering (mesh generators number spatially coherent cells together)"""
    return (np.arange(n_cells, dtype=np.int64) * world) // max(n_cells, 1)


def partition_mesh(lm, owner, rank, chunk=512, world=None):
    """(LocalMesh of `rank`, gid) from the one-rank LocalMesh `lm` and owner[cell] in 0 .. world - 1.

    gid: the one-rank index of every local owned-or-ghost cell (initial states, regions and results are gathered with it;
    also LocalMesh.extras["prim_gid"]).  Preconditioner subdomains: the one-rank mesh's subdomains cut at the ownership
    boundaries, then chunks of at most `chunk` consecutive owned cells."""
    if lm.n_halo:
        raise ValueError("partition_mesh wants a one-rank mesh")
    owner = np.asarray(owner, dtype=np.int64)
    N, NB = lm.n_owned, lm.n_bc
    if owner.size != N:
        raise ValueError("one owner per cell")
    # `world` = the communicator's size (advisor, round 5: inferred from the owner array alone it is wrong for an array that
    # leaves the last ranks empty, and an empty rank fails far downstream); every rank must own a cell
    if world is None:
        world = int(owner.max()) + 1
    if owner.min() < 0 or owner.max() >= world:
        raise ValueError("owner outside 0 .. %d" % (world - 1))
    if not 0 <= rank < world:
        raise ValueError("rank %d of %d" % (rank, world))
    empty = np.setdiff1d(np.arange(world), np.unique(owner))
    if empty.size:
        raise ValueError("rank(s) %s own no cell (%d cells on %d ranks)" % (empty[:8].tolist(), N, world))
    fc = np.asarray(lm.face_cells, dtype=np.int64).reshape(-1, 2)
    own_f = np.where(fc < N, owner[np.clip(fc, 0, N - 1)], -1)          # owner of each face cell, -1: boundary cell
    keep = (own_f == rank).any(axis=1)
    fck = fc[keep]
    ofk = own_f[keep]
    mine = np.nonzero(owner == rank)[0]
    # ghosts: other ranks' cells across a kept face; grouped by owner, ascending global index inside a group
    other = (ofk != rank) & (ofk >= 0)
    ghosts = np.unique(fck[other])
    gowner = owner[ghosts]
    order = np.lexsort((ghosts, gowner))
    ghosts, gowner = ghosts[order], gowner[order]
    nbr = np.unique(gowner)
    recv_ptr = np.concatenate([[0], np.cumsum([(gowner == q).sum() for q in nbr])]).astype(np.int32)
    # Dirichlet boundary cells hanging on my cells
    bcs = np.unique(fck[ofk < 0])
    n_o, n_h, n_b = mine.size, ghosts.size, bcs.size
    loc = np.full(N + NB, -1, dtype=np.int64)
    loc[mine] = np.arange(n_o)
    loc[ghosts] = n_o + np.arange(n_h)
    loc[bcs] = n_o + n_h + np.arange(n_b)
    m = copy.copy(lm)
    m.extras = dict(lm.extras)
    m.part, m.rank = (world, 1, 1), rank
    m.n_owned, m.n_halo, m.n_bc = int(n_o), int(n_h), int(n_b)
    m.face_cells = loc[fck].astype(np.int32)
    assert (m.face_cells >= 0).all()
    m.face_geom = np.asarray(lm.face_geom).reshape(-1, 12)[keep].copy()
    m.n_faces = int(keep.sum())
    sel = np.concatenate([mine, ghosts, bcs])
    m.cell_geom = np.asarray(lm.cell_geom)[sel].copy()
    m.rock = np.asarray(lm.rock)[sel].copy()
    if n_b:
        m.bc_primary = np.asarray(lm.bc_primary)[bcs - N].copy()
        m.bc_region = np.asarray(lm.bc_region)[bcs - N].copy()
        if "bc_spec" in lm.extras:
            m.extras["bc_spec"] = np.asarray(lm.extras["bc_spec"])[bcs - N].copy()
    else:
        m.bc_primary, m.bc_region = None, None
    # send lists: for neighbour q, my cells across a kept face from q's cells -- ascending global index, which is the
    # order of q's ghost block for me
    send_idx, send_ptr = [], [0]
    for q in nbr:
        a = (ofk[:, 0] == rank) & (ofk[:, 1] == q)
        b = (ofk[:, 1] == rank) & (ofk[:, 0] == q)
        cells = np.unique(np.concatenate([fck[a, 0], fck[b, 1]]))
        send_idx.append(loc[cells])
        send_ptr.append(send_ptr[-1] + cells.size)
    m.nbr_ranks = nbr.astype(np.int32)
    m.send_ptr = np.asarray(send_ptr, dtype=np.int32)
    m.send_idx = (np.concatenate(send_idx) if send_idx else np.zeros(0)).astype(np.int32)
    m.recv_ptr = recv_ptr
    # subdomains: the one-rank mesh's, cut where ownership changes, and no longer than `chunk`
    cuts = {0, n_o}
    if lm.sub_ptr is not None:
        sub_of = np.searchsorted(np.asarray(lm.sub_ptr), mine, side="right") - 1
        cuts.update((np.nonzero(np.diff(sub_of))[0] + 1).tolist())
    cuts.update((np.nonzero(np.diff(mine) != 1)[0] + 1).tolist())       # not consecutive in the one-rank numbering
    pts = sorted(cuts)
    sub = [0]
    for a, b in zip(pts[:-1], pts[1:]):
        k = a
        while k < b:
            k = min(k + chunk, b)
            sub.append(k)
    m.sub_ptr = np.asarray(sub, dtype=np.int32)
    gid = np.concatenate([mine, ghosts])
    m.owned_gid = mine.copy()
    m.owned_ijk = np.asarray(lm.owned_ijk)[mine].copy() if getattr(lm, "owned_ijk", None) is not None else None
    m.extras["prim_gid"] = gid
    for key in ("minc_level", "minc_parent"):
        if key in lm.extras:
            m.extras[key] = np.asarray(lm.extras[key])[mine].copy()
    # sources of my cells, with their index in the one-rank list (wai_set_source_global_index)
    if lm.n_src:
        sc = np.asarray(lm.src_cell)
        pick = np.nonzero(owner[sc] == rank)[0]
        m.n_src = int(pick.size)
        m.src_cell = loc[sc[pick]].astype(np.int32)
        m.src_rate = np.asarray(lm.src_rate)[pick].copy()
        m.src_enthalpy = np.asarray(lm.src_enthalpy)[pick].copy()
        m.src_component = np.asarray(lm.src_component)[pick].copy()
        m.extras["src_global_index"] = pick.astype(np.int32)
    return m, gid
