"""Synthetic structured finite-volume meshes in the flat-array form the hot path consumes.

This replaces, for synthetic inputs, what the reference builds with PETSc DMPlex in
src/mesh.F90 (out of scope as code): what matters to the Newton-step path
are its *outputs* -- the DMPlex-local arrays

* cell geometry  [centroid(3), volume]                       (src/cell.F90:54-61,85-96)
* face geometry  [area, d1, d2, d12, n(3), g.n, centroid(3), dir]  (src/face.F90:67-76,119-135)
* rock           [k1,k2,k3, wet, dry, porosity, density, cp]  (src/rock.F90:56-65,97-112)
* face -> (cell1, cell2) support, normal pointing 1 -> 2     (src/mesh.F90:462-579)
* Dirichlet boundary ghost cells appended after the interior cells, zero volume,
  d = (d1, 0), d12 = d1, rock copied from the interior cell    (src/mesh.F90:583-664,1189-1202)
* one-cell overlap between partitions                          (src/mesh.F90:40,160)

Cell numbering is *brick-major*: the owned block of a rank is tiled by bricks (bx,by,bz) and
cells are numbered brick after brick, x-fastest inside a brick.  A brick is one block-Jacobi /
ILU(0) subdomain of the preconditioner (the reference's PCBJACOBI/PCASM with one block per MPI
rank, src/timestepper.F90:1668-1669; here one block per brick so a subdomain fits a CU's LDS).

Local cell order on a rank:  [owned | halo (other ranks' cells, grouped by neighbour) | bc].
"""
from dataclasses import dataclass, field

import re

import numpy as np

GRAVITY = 9.8  # default 3-D gravity (0,0,-9.8): src/flow_simulation.F90:833-846


def _splits(n, parts):
    """Balanced contiguous split of range(n) into `parts` pieces: boundaries array."""
    base, extra = divmod(n, parts)
    sizes = np.full(parts, base, dtype=np.int64)
    sizes[:extra] += 1
    return np.concatenate([[0], np.cumsum(sizes)])


class _Axis:
    """Per-axis decomposition.  Ranks get a balanced contiguous range of cells; every rank's range
    is tiled with bricks starting at its lower end (the last brick of a rank may be ragged), so
    preconditioner bricks never straddle ranks and the load stays balanced whatever the brick."""

    def __init__(self, n, parts, brick, balanced=False):
        # balanced: a rank's range is cut into ceil(range / brick) bricks of (nearly) EQUAL size instead of full bricks and
        # one remainder -- 216 cells in bricks of 16: fourteen bricks of 15 or 16 instead of thirteen of 16 and one of 8.  A
        # half or quarter brick holds a CU slot of the fused kernel for almost as long as a full one
        self.n, self.parts = n, parts
        csplit = _splits(n, parts)                      # balanced cell ranges
        self.rank_lo = csplit[:-1].copy()
        self.rank_hi = csplit[1:].copy()
        edges, bsplit = [], [0]
        for r in range(parts):
            lo, hi = int(csplit[r]), int(csplit[r + 1])
            if balanced and hi > lo:
                nb = -(-(hi - lo) // brick)
                e = [lo + int(v) for v in _splits(hi - lo, nb)[:-1]]
            else:
                e = list(range(lo, hi, brick))
            edges += e
            bsplit.append(len(edges))
        edges.append(n)
        edges = np.array(edges, dtype=np.int64)
        bsplit = np.array(bsplit, dtype=np.int64)
        c = np.arange(n)
        self.brick_of = np.searchsorted(edges, c, side="right") - 1
        self.rank_of = np.searchsorted(csplit, c, side="right") - 1
        self.brick_in_rank = self.brick_of - bsplit[self.rank_of]
        self.off = c - edges[self.brick_of]
        self.bsize = (edges[1:] - edges[:-1])[self.brick_of]
        self.nbricks_rank = bsplit[1:] - bsplit[:-1]
        self.bsplit = bsplit
        self.edges = edges


@dataclass
class LocalMesh:
    dims: tuple
    spacing: tuple
    part: tuple
    rank: int
    brick: tuple
    n_owned: int = 0
    n_halo: int = 0
    n_bc: int = 0
    n_faces: int = 0
    face_cells: np.ndarray = None
    face_geom: np.ndarray = None
    cell_geom: np.ndarray = None
    rock: np.ndarray = None
    bc_primary: np.ndarray = None
    bc_region: np.ndarray = None
    sub_ptr: np.ndarray = None
    owned_gid: np.ndarray = None      # natural global index (k*ny + j)*nx + i of each owned cell
    owned_ijk: np.ndarray = None
    nbr_ranks: np.ndarray = None      # halo description: neighbour ranks,
    send_ptr: np.ndarray = None       # send_idx[send_ptr[q]:send_ptr[q+1]] = owned cells sent to q
    send_idx: np.ndarray = None
    recv_ptr: np.ndarray = None       # halo cells n_owned+recv_ptr[q] .. received from q
    n_src: int = 0
    src_cell: np.ndarray = None
    src_rate: np.ndarray = None
    src_enthalpy: np.ndarray = None
    src_component: np.ndarray = None
    n_global: int = 0
    extras: dict = field(default_factory=dict)

    @property
    def n_prim(self):
        return self.n_owned + self.n_halo

    @property
    def n_local(self):
        return self.n_owned + self.n_halo + self.n_bc


class StructuredGrid:
    """Global description of an nx*ny*nz box split over px*py*pz ranks."""

    def __init__(self, dims, spacing=(10.0, 10.0, 10.0), part=(1, 1, 1), brick=(8, 8, 8),
                 order="hyperplane", brick_order="x", balanced_bricks=False):
        # order: numbering of the cells inside a brick.  "hyperplane" sorts them by i+j+k (ties in
        # natural order) = by dependency level of the brick's ILU(0) factors, so storage order is
        # level order and a wavefront owns whole consecutive levels; "natural" is x-fastest.
        self.order = order
        # brick_order: which way the bricks of a rank are numbered: "x" x fastest (default), "z" with the brick above /
        # below following directly.  The block-Jacobi bricks are independent, so the preconditioner does not depend on
        # it; it only moves the out-of-brick neighbours in memory.  MEASURED at 216^3 (rocprofv3, one box, round 3): "z"
        # k_pc_park 621 against 631 us, but k_jacobian_park 10.87 against 10.57 ms and k_residual 2.73 against 2.66 ms --
        # the assembly sweeps' re-reads are a capacity problem (a brick's 31 field planes of own + neighbour lines, ~160 KB,
        # against 128 KB of L2 per CU), not a distance problem; not adopted.
        # "tile" / "tileN" (round 6): x fastest, then y inside strips of N (default 4) brick rows, then z, then the strips.
        # A fused launch keeps ~96 bricks resident per XCD (3 per CU) and works through its eighth of the numbering in
        # order, so a neighbour brick's vector entries are L2 hits when that brick is within about a hundred positions.
        # With 14 x 14 x 108 bricks of 16 x 16 x 2 cells (216^3) "x" puts the bricks above / below 196 positions away --
        # and those are the expensive neighbours: with the cells of a brick in level order their 256 rows touch every
        # line of both bricks (16 KB per gathered vector and brick, against 4 KB for each x / y face).  In strips of 4
        # brick rows the distances are x 1, y 14, z 56, and only a quarter of the y links cross a strip.
        # "tileAxB": columns of A x B bricks -- x fastest inside the column's A, then its B rows, then z through all layers,
        # then the next column (x, then y): the bricks above / below are A * B positions away.  ("tileN" = all of x, N rows.)
        m_ = re.fullmatch(r"tile(\d*)(?:x(\d+))?", brick_order) if isinstance(brick_order, str) else None
        if brick_order not in ("z", "x") and not m_:
            raise ValueError("brick_order 'z', 'x', 'tile[N]' or 'tileAxB'")
        self.brick_order = "tile" if m_ else brick_order
        if m_ and m_.group(2):
            self.brick_tile_x, self.brick_tile_y = int(m_.group(1) or 0), int(m_.group(2))
            if self.brick_tile_x < 1:
                raise ValueError("brick_order tileAxB: A >= 1")
        else:
            self.brick_tile_x, self.brick_tile_y = 0, int(m_.group(1)) if m_ and m_.group(1) else 4   # 0: the whole row
        if self.brick_tile_y < 1:
            raise ValueError("brick_order tileN / tileAxB: N, B >= 1")
        self.dims = tuple(int(v) for v in dims)
        self.spacing = tuple(float(v) for v in spacing)
        self.part = tuple(int(v) for v in part)
        self.brick = tuple(int(v) for v in brick)
        self.balanced_bricks = bool(balanced_bricks)
        self.ax = [_Axis(self.dims[a], self.part[a], self.brick[a], self.balanced_bricks) for a in range(3)]
        self.nranks = self.part[0] * self.part[1] * self.part[2]
        self.n_global = self.dims[0] * self.dims[1] * self.dims[2]

    def rank_coords(self, rank):
        px, py, _ = self.part
        return rank % px, (rank // px) % py, rank // (px * py)

    def rank_id(self, rx, ry, rz):
        px, py, _ = self.part
        return (rz * py + ry) * px + rx

    def owner(self, i, j, k):
        return self.rank_id(self.ax[0].rank_of[i], self.ax[1].rank_of[j], self.ax[2].rank_of[k])

    def _brick_starts(self, rc):
        """Start offset of every brick of rank `rc` (brick-natural order, x fastest)."""
        sizes = []
        for a in range(3):
            ax = self.ax[a]
            b0, b1 = ax.bsplit[rc[a]], ax.bsplit[rc[a] + 1]
            sizes.append((ax.edges[b0 + 1:b1 + 1] - ax.edges[b0:b1]))
        vol = sizes[2][:, None, None] * sizes[1][None, :, None] * sizes[0][None, None, :]
        nbz, nby, nbx = vol.shape
        bz, by, bx = np.meshgrid(np.arange(nbz), np.arange(nby), np.arange(nbx), indexing="ij")
        if self.brick_order == "z":     # numbered (bx, by, bz) with bz fastest
            pos = (bx * nby + by) * nbz + bz
        elif self.brick_order == "tile":   # x fastest, y inside its strip of brick rows, z, then the strips
            ty, tx = self.brick_tile_y, (self.brick_tile_x or nbx)
            rows_before = np.minimum((by // ty) * ty, nby)                  # brick rows in the strips before this one
            rows_here = np.minimum(ty, nby - (by // ty) * ty)                # rows of this strip (the last one may be short)
            cols_before = np.minimum((bx // tx) * tx, nbx)                  # inside the strip: the columns before this one
            cols_here = np.minimum(tx, nbx - (bx // tx) * tx)
            pos = (rows_before * nbx + rows_here * cols_before) * nbz + (bz * rows_here + by % ty) * cols_here + bx % tx
        else:
            pos = (bz * nby + by) * nbx + bx
        byvol = np.empty(vol.size, dtype=np.int64)
        byvol[pos.ravel()] = vol.ravel()
        starts = np.concatenate([[0], np.cumsum(byvol)])
        return starts, pos

    def _brick_index(self, bz, by, bx, pos):
        """number of brick (bx, by, bz) of a rank; pos as returned by _brick_starts"""
        return pos[bz, by, bx]

    def local_id(self, rank, i, j, k):
        """Local (owned) index on `rank` of global cells (i,j,k) that rank owns."""
        rc = self.rank_coords(rank)
        starts, shape = self._brick_starts(rc)
        ax, ay, az = self.ax
        b = self._brick_index(az.brick_in_rank[k], ay.brick_in_rank[j], ax.brick_in_rank[i], shape)
        within = (az.off[k] * ay.bsize[j] + ay.off[j]) * ax.bsize[i] + ax.off[i]
        if self.order == "hyperplane":
            # position in the brick's level order: cells sorted by (ox + oy + oz, natural index),
            # the numbering local_mesh gives the brick (few cells are asked for: sources)
            within = np.asarray(within).copy()
            for q in range(within.size):
                sx, sy, sz = int(ax.bsize[i[q]]), int(ay.bsize[j[q]]), int(az.bsize[k[q]])
                oz, oy, ox = np.meshgrid(np.arange(sz), np.arange(sy), np.arange(sx), indexing="ij")
                nat = ((oz * sy + oy) * sx + ox).ravel()
                key = (oz + oy + ox).ravel() * nat.size + nat
                mine = (az.off[k[q]] + ay.off[j[q]] + ax.off[i[q]]) * nat.size + within[q]
                within[q] = np.count_nonzero(key < mine)
        return (starts[b] + within).astype(np.int64)

    def natural_id(self, i, j, k):
        nx, ny, _ = self.dims
        return (np.asarray(k, dtype=np.int64) * ny + j) * nx + i

    def _local_mesh_minc(self, rank, rock_fn, top_bc, sources, minc):
        base = self.local_mesh(rank, rock_fn=rock_fn, top_bc=top_bc, sources=sources)
        return _add_minc(self, base, minc)

    # -------------------------------------------------------------------------------------
    def local_mesh(self, rank=0, rock_fn=None, top_bc=None, sources=None, minc=None):
        """Build the flat arrays of one rank.

        rock_fn(gid) -> (n, 8) rock records for natural cell ids; top_bc = (primary, region)
        puts Dirichlet ghost cells on the top (k = 0) faces; sources = list of dicts
        {ijk, rate, enthalpy, component} in global coordinates.  minc = dict(geometry=MincGeometry,
        matrix_rock=(8,) record) adds the MINC matrix cells of every owned fracture cell
        (src/mesh.F90:3026-3186): inside a brick the fracture cells come first, then level 1, ...;
        matrix cell of level m is connected to level m-1 by a face with area V*connection_area(m),
        distances (cd(m), cd(m+1)), zero normal and gravity term, permeability direction 1.
        """
        if minc is not None:
            return self._local_mesh_minc(rank, rock_fn, top_bc, sources, minc)
        nx, ny, nz = self.dims
        dx, dy, dz = self.spacing
        rc = self.rank_coords(rank)
        lo = [self.ax[a].rank_lo[rc[a]] for a in range(3)]
        hi = [self.ax[a].rank_hi[rc[a]] for a in range(3)]
        m = LocalMesh(dims=self.dims, spacing=self.spacing, part=self.part, rank=rank,
                      brick=self.brick, n_global=self.n_global)
        # natural-order (k,j,i) table of local ids over the owned block, padded by one cell for
        # the halo slabs; everything below is slicing / broadcasting (no per-cell gathers)
        ax, ay, az = self.ax
        ri = np.arange(lo[0], hi[0]); rj = np.arange(lo[1], hi[1]); rk = np.arange(lo[2], hi[2])
        starts, bshape = self._brick_starts(rc)
        bidx = self._brick_index(az.brick_in_rank[rk][:, None, None], ay.brick_in_rank[rj][None, :, None],
                                 ax.brick_in_rank[ri][None, None, :], bshape)
        within = ((az.off[rk][:, None, None] * ay.bsize[rj][None, :, None] + ay.off[rj][None, :, None])
                  * ax.bsize[ri][None, None, :] + ax.off[ri][None, None, :])
        if self.order == "hyperplane":
            ux, ix_id = np.unique(ax.bsize[ri], return_inverse=True)
            uy, iy_id = np.unique(ay.bsize[rj], return_inverse=True)
            uz, iz_id = np.unique(az.bsize[rk], return_inverse=True)
            maxvol = int(ux.max() * uy.max() * uz.max())
            tables = np.zeros((len(uz) * len(uy) * len(ux), maxvol), dtype=np.int64)
            for a_, sz_ in enumerate(uz):
                for b_, sy_ in enumerate(uy):
                    for c_, sx_ in enumerate(ux):
                        oz_, oy_, ox_ = np.meshgrid(np.arange(sz_), np.arange(sy_), np.arange(sx_), indexing="ij")
                        nat = ((oz_ * sy_ + oy_) * sx_ + ox_).ravel()
                        key = (oz_ + oy_ + ox_).ravel() * nat.size + nat
                        hrank = np.empty(nat.size, dtype=np.int64)
                        hrank[np.argsort(key, kind="stable")] = np.arange(nat.size)
                        tables[(a_ * len(uy) + b_) * len(ux) + c_, nat] = hrank
            sid3 = ((iz_id[:, None, None] * len(uy) + iy_id[None, :, None]) * len(ux) + ix_id[None, None, :])
            within = tables[sid3, within]
            del sid3
        lid3 = (starts[bidx] + within).astype(np.int64)
        del bidx, within
        nzl, nyl, nxl = lid3.shape
        n_owned = lid3.size
        nat_local = np.empty(n_owned, dtype=np.int64)   # local id -> natural index inside the block
        nat_local[lid3.ravel()] = np.arange(n_owned)
        ok, rem = np.divmod(nat_local, nyl * nxl)
        oj, oi = np.divmod(rem, nxl)
        del rem, nat_local
        oi += lo[0]; oj += lo[1]; ok += lo[2]
        m.n_owned = n_owned
        m.owned_ijk = np.stack([oi, oj, ok], axis=1).astype(np.int32)
        m.owned_gid = self.natural_id(oi, oj, ok)
        m.sub_ptr = starts.astype(np.int32)
        lut = np.full((nzl + 2, nyl + 2, nxl + 2), -1, dtype=np.int64)
        lut[1:-1, 1:-1, 1:-1] = lid3
        del lid3
        # halo slabs, neighbour order -x,+x,-y,+y,-z,+z; slab cells in natural order
        nbr_ranks, send_idx, send_ptr, recv_ptr = [], [], [0], [0]
        halo_ijk = []
        for a in range(3):
            for sgn in (-1, 1):
                nrc = list(rc)
                nrc[a] += sgn
                if nrc[a] < 0 or nrc[a] >= self.part[a]:
                    continue
                nrank = self.rank_id(*nrc)
                rng = [np.arange(lo[b], hi[b]) for b in range(3)]
                theirs = [r.copy() for r in rng]
                theirs[a] = np.array([lo[a] - 1 if sgn < 0 else hi[a]])
                TK, TJ, TI = np.meshgrid(theirs[2], theirs[1], theirs[0], indexing="ij")
                ti, tj, tk = TI.ravel(), TJ.ravel(), TK.ravel()
                nh = ti.size
                base = n_owned + recv_ptr[-1]
                # slices of the padded table: the slab itself and my boundary layer next to it
                sl_t = [slice(1, -1), slice(1, -1), slice(1, -1)]   # (k, j, i)
                sl_m = [slice(1, -1), slice(1, -1), slice(1, -1)]
                pos_t = 0 if sgn < 0 else (hi[a] - lo[a] + 1)
                pos_m = 1 if sgn < 0 else (hi[a] - lo[a])
                sl_t[2 - a] = slice(pos_t, pos_t + 1)
                sl_m[2 - a] = slice(pos_m, pos_m + 1)
                lut[tuple(sl_t)] = (base + np.arange(nh)).reshape(lut[tuple(sl_t)].shape)
                halo_ijk.append(np.stack([ti, tj, tk], axis=1))
                nbr_ranks.append(nrank)
                recv_ptr.append(recv_ptr[-1] + nh)
                send_idx.append(lut[tuple(sl_m)].ravel().copy())
                send_ptr.append(send_ptr[-1] + nh)
        m.n_halo = recv_ptr[-1]
        m.nbr_ranks = np.array(nbr_ranks, dtype=np.int32)
        m.send_ptr = np.array(send_ptr, dtype=np.int32)
        m.recv_ptr = np.array(recv_ptr, dtype=np.int32)
        m.send_idx = (np.concatenate(send_idx) if send_idx else np.zeros(0)).astype(np.int32)
        hijk = np.concatenate(halo_ijk) if halo_ijk else np.zeros((0, 3), dtype=np.int64)
        n_prim = n_owned + m.n_halo
        pi = np.concatenate([oi, hijk[:, 0]]).astype(np.int64)
        pj = np.concatenate([oj, hijk[:, 1]]).astype(np.int64)
        pk = np.concatenate([ok, hijk[:, 2]]).astype(np.int64)
        m.extras["prim_ijk"] = np.stack([pi, pj, pk], axis=1)
        m.extras["prim_gid"] = self.natural_id(pi, pj, pk)

        # interior faces: every face with at least one owned cell, in natural order of the first
        # cell; c1 = lower index along the axis, normal = +axis for x,y; for z the cell with
        # smaller k is the *upper* one (k = 0 is the top layer), normal = (0,0,-1), g.n = +9.8
        fc, fg = [], []
        area = (dy * dz, dx * dz, dx * dy)
        dist = (dx, dy, dz)
        for a in range(3):
            # padded-table range of the first cell of each pair along axis a
            first_lo = 0 if lo[a] > 0 else 1
            first_hi = (hi[a] - lo[a] + 1) if hi[a] < self.dims[a] else (hi[a] - lo[a])
            if first_hi <= first_lo:
                continue
            s1 = [slice(1, -1), slice(1, -1), slice(1, -1)]
            s2 = [slice(1, -1), slice(1, -1), slice(1, -1)]
            s1[2 - a] = slice(first_lo, first_hi)
            s2[2 - a] = slice(first_lo + 1, first_hi + 1)
            c1 = lut[tuple(s1)]
            c2 = lut[tuple(s2)]
            shp3 = c1.shape
            c1 = c1.ravel(); c2 = c2.ravel()
            # coordinates of the first cell
            kk = np.arange(shp3[0]) + (lo[2] if a != 2 else lo[2] + first_lo - 1)
            jj = np.arange(shp3[1]) + (lo[1] if a != 1 else lo[1] + first_lo - 1)
            ii = np.arange(shp3[2]) + (lo[0] if a != 0 else lo[0] + first_lo - 1)
            nfa = c1.size
            g = np.zeros((nfa, 12))
            g[:, 0] = area[a]
            g[:, 1] = 0.5 * dist[a]
            g[:, 2] = 0.5 * dist[a]
            g[:, 3] = dist[a]
            nsign = 1.0 if a < 2 else -1.0
            g[:, 4 + a] = nsign
            g[:, 7] = GRAVITY if a == 2 else 0.0   # g = (0,0,-9.8), n = (0,0,-1)
            g3 = g.reshape(shp3 + (12,))
            g3[..., 8] = ((ii + 0.5) * dx)[None, None, :]
            g3[..., 9] = ((jj + 0.5) * dy)[None, :, None]
            g3[..., 10] = (-(kk + 0.5) * dz)[:, None, None]
            g3[..., 8 + a] += 0.5 * dist[a] * nsign
            g[:, 11] = a + 1
            fc.append(np.stack([c1, c2], axis=1))
            fg.append(g)
        # Dirichlet boundary ghost cells on the top faces of owned k = 0 cells
        bc_cells = np.zeros(0, dtype=np.int64)
        if top_bc is not None and lo[2] == 0:
            top = np.nonzero(ok == 0)[0]
            bc_cells = top
            nb = top.size
            ghost = n_prim + np.arange(nb)
            g = np.zeros((nb, 12))
            g[:, 0] = area[2]
            g[:, 1] = 0.5 * dz
            g[:, 2] = 0.0
            g[:, 3] = 0.5 * dz
            g[:, 6] = 1.0
            g[:, 7] = -GRAVITY
            g[:, 8] = (oi[top] + 0.5) * dx
            g[:, 9] = (oj[top] + 0.5) * dy
            g[:, 10] = 0.0
            g[:, 11] = 3
            fc.append(np.stack([top, ghost], axis=1))
            fg.append(g)
            prim, region = top_bc
            m.bc_primary = np.tile(np.asarray(prim, dtype=np.float64), (nb, 1))
            m.bc_region = np.full(nb, int(region), dtype=np.int32)
        m.n_bc = bc_cells.size
        fc = [a for a in fc if a.shape[0]] or [np.zeros((0, 2), dtype=np.int64)]
        fg = [a for a in fg if a.shape[0]] or [np.zeros((0, 12))]
        m.face_cells = np.concatenate(fc).astype(np.int32)
        m.face_geom = np.concatenate(fg)
        m.n_faces = m.face_cells.shape[0]
        # cell geometry
        n_local = n_prim + m.n_bc
        cg = np.zeros((n_local, 4))
        cg[:n_prim, 0] = (pi + 0.5) * dx
        cg[:n_prim, 1] = (pj + 0.5) * dy
        cg[:n_prim, 2] = -(pk + 0.5) * dz
        cg[:n_prim, 3] = dx * dy * dz
        if m.n_bc:
            cg[n_prim:, 0:3] = m.face_geom[-m.n_bc:, 8:11]
            cg[n_prim:, 3] = 0.0
        m.cell_geom = cg
        # rock
        rock = np.zeros((n_local, 8))
        gids = m.extras["prim_gid"]
        if rock_fn is None:
            rock[:n_prim] = default_rock(gids.size)
        else:
            rock[:n_prim] = rock_fn(gids)
        if m.n_bc:
            rock[n_prim:] = rock[bc_cells]
        m.rock = rock
        # sources owned by this rank
        if sources:
            sc, sr, se, sk = [], [], [], []
            for s in sources:
                i, j, k = s["ijk"]
                if self.owner(i, j, k) == rank:
                    sc.append(int(self.local_id(rank, np.array([i]), np.array([j]), np.array([k]))[0]))
                    sr.append(s["rate"])
                    se.append(s.get("enthalpy", 0.0))
                    sk.append(s.get("component", 0))
            m.n_src = len(sc)
            m.src_cell = np.array(sc, dtype=np.int32)
            m.src_rate = np.array(sr, dtype=np.float64)
            m.src_enthalpy = np.array(se, dtype=np.float64)
            m.src_component = np.array(sk, dtype=np.int32)
        return m


class MincGeometry:
    """MINC 'nested cube' geometry (src/minc.F90:393-548): proximity function, its derivative,
    innermost connection distance and the per-level connection areas / distances of
    minc_setup_geometry.  volumes = [fracture, matrix level 1, ...] volume fractions."""

    def __init__(self, volumes, spacing, fracture_connection_distance=0.0):
        self.volume = np.asarray(volumes, dtype=np.float64)
        self.volume = self.volume / self.volume.sum()
        self.spacing = np.atleast_1d(np.asarray(spacing, dtype=np.float64))
        self.num_planes = self.spacing.size
        self.num_levels = self.volume.size - 1
        self.fracture_connection_distance = fracture_connection_distance
        self._setup()

    def proximity(self, d):                       # minc.F90:393-411
        fout = 1.0 - 2.0 * d / self.spacing
        return 1.0 if np.any(fout < 0.0) else 1.0 - float(np.prod(fout))

    def proximity_derivative(self, d):            # minc.F90:415-434
        fout = 1.0 - 2.0 * d / self.spacing
        if np.any(fout < 0.0):
            return 0.0
        excl = np.array([np.prod(np.delete(fout, i)) for i in range(fout.size)])
        return 2.0 * float(np.sum(excl / self.spacing))

    def inner_connection_distance(self, x):       # minc.F90:438-461 (Pruess 1983, GMINC)
        u = self.spacing - 2.0 * x
        if self.num_planes == 1:
            return u[0] / 6.0
        if self.num_planes == 2:
            return 0.25 * np.prod(u) / np.sum(u)
        pair = sum(u[i] * u[(i + 1) % 3] for i in range(3))
        return 0.3 * np.prod(u) / pair

    def _setup(self):                             # minc.F90:465-525
        from scipy.optimize import brentq
        vmatrix = 1.0 - self.volume[0]
        volsum = np.cumsum(self.volume[1:]) / vmatrix
        nl = self.num_levels
        self.connection_distance = np.zeros(nl + 1)
        self.connection_area = np.zeros(nl)
        x = 0.0
        self.connection_distance[0] = self.fracture_connection_distance
        self.connection_area[0] = vmatrix * self.proximity_derivative(x)
        xr = self.volume[1] / self.connection_area[0]
        for i in range(nl - 1):
            xl = x
            f = lambda xx, v=volsum[i]: self.proximity(xx) - v
            while f(xr) < 0.0:
                xr *= 2.0
            x = brentq(f, xl, xr, xtol=1e-12, rtol=1e-14)
            self.connection_distance[i + 1] = 0.5 * (x - xl)
            self.connection_area[i + 1] = vmatrix * self.proximity_derivative(x)
        self.connection_distance[nl] = self.inner_connection_distance(x)


def _add_minc(grid, base, minc):
    """Rebuild a fracture-only LocalMesh with MINC matrix cells inserted brick by brick."""
    geo, mrock = minc["geometry"], np.asarray(minc["matrix_rock"], dtype=np.float64)
    nl = geo.num_levels
    no = base.n_owned
    sp = base.sub_ptr.astype(np.int64)
    nsub = sp.size - 1
    bsize = np.diff(sp)
    # new index of fracture cell i and of its level-m matrix cell
    sub_of = np.repeat(np.arange(nsub), bsize)
    new_start = sp * (nl + 1)
    within = np.arange(no) - sp[sub_of]
    frac_new = new_start[sub_of] + within
    level_new = [new_start[sub_of] + bsize[sub_of] * m + within for m in range(1, nl + 1)]
    n_owned_new = no * (nl + 1)

    def remap(idx):
        idx = np.asarray(idx, dtype=np.int64)
        out = np.where(idx < no, 0, idx + no * nl)
        own = idx < no
        out[own] = frac_new[idx[own]]
        return out
    m = LocalMesh(dims=base.dims, spacing=base.spacing, part=base.part, rank=base.rank, brick=base.brick,
                  n_global=base.n_global)
    m.n_owned, m.n_halo, m.n_bc = n_owned_new, base.n_halo, base.n_bc
    n_local = m.n_local
    # cells
    cg = np.zeros((n_local, 4))
    rock = np.zeros((n_local, 8))
    cg[remap(np.arange(base.n_local))] = base.cell_geom
    rock[remap(np.arange(base.n_local))] = base.rock
    vol0 = base.cell_geom[:no, 3]
    cg[frac_new, 3] = vol0 * geo.volume[0]
    for mlev in range(1, nl + 1):
        cg[level_new[mlev - 1], 0:3] = base.cell_geom[:no, 0:3]
        cg[level_new[mlev - 1], 3] = vol0 * geo.volume[mlev]
        rock[level_new[mlev - 1]] = mrock
    # halo fracture cells keep the reduced fracture volume too (only used through faces of owned cells)
    cg[n_owned_new:n_owned_new + base.n_halo, 3] *= geo.volume[0]
    m.cell_geom, m.rock = cg, rock
    # faces: original ones remapped, then MINC faces level by level
    fc = [remap(base.face_cells.ravel()).reshape(-1, 2)]
    fg = [base.face_geom]
    for mlev in range(1, nl + 1):
        c1 = frac_new if mlev == 1 else level_new[mlev - 2]
        c2 = level_new[mlev - 1]
        g = np.zeros((no, 12))
        g[:, 0] = vol0 * geo.connection_area[mlev - 1]
        g[:, 1] = geo.connection_distance[mlev - 1]
        g[:, 2] = geo.connection_distance[mlev]
        g[:, 3] = g[:, 1] + g[:, 2]
        g[:, 8:11] = base.cell_geom[:no, 0:3]
        g[:, 11] = 1.0
        fc.append(np.stack([c1, c2], axis=1))
        fg.append(g)
    m.face_cells = np.concatenate(fc).astype(np.int32)
    m.face_geom = np.concatenate(fg)
    m.n_faces = m.face_cells.shape[0]
    m.sub_ptr = (sp * (nl + 1)).astype(np.int32)
    m.bc_primary, m.bc_region = base.bc_primary, base.bc_region
    m.owned_gid = np.full(n_owned_new, -1, dtype=np.int64)
    m.owned_gid[frac_new] = base.owned_gid
    m.owned_ijk = np.zeros((n_owned_new, 3), dtype=np.int32)
    m.owned_ijk[frac_new] = base.owned_ijk
    for mlev in range(1, nl + 1):
        m.owned_ijk[level_new[mlev - 1]] = base.owned_ijk
    m.extras["minc_level"] = np.zeros(n_owned_new, dtype=np.int32)
    for mlev in range(1, nl + 1):
        m.extras["minc_level"][level_new[mlev - 1]] = mlev
    m.extras["fracture_index"] = frac_new
    pijk = np.concatenate([m.owned_ijk.astype(np.int64), base.extras["prim_ijk"][no:]])
    m.extras["prim_ijk"] = pijk
    m.nbr_ranks, m.send_ptr, m.recv_ptr = base.nbr_ranks, base.send_ptr, base.recv_ptr
    m.send_idx = frac_new[base.send_idx].astype(np.int32) if base.send_idx is not None and base.send_idx.size else base.send_idx
    m.n_src = base.n_src
    if base.n_src:
        m.src_cell = frac_new[base.src_cell].astype(np.int32)
        m.src_rate, m.src_enthalpy, m.src_component = base.src_rate, base.src_enthalpy, base.src_component
    return m


def add_minc_zones(base, zones):
    """MINC matrix cells for part of a single-rank mesh (`"mesh": {"minc": ...}` of an input file,
    src/mesh.F90:3026-3186, src/minc.F90:58-374).  zones: [dict(cells=original cell indices,
    geometry=MincGeometry, matrix_rock=(8,), fracture_rock=(8,) or None)].  A zone cell becomes the
    fracture cell (volume times the fracture fraction, optionally the fracture rock type); its matrix
    cells of level m are connected in a chain by faces of area V*connection_area(m), distances
    (cd(m), cd(m+1)), zero normal and gravity term, permeability direction 1.  Storage order: inside
    each preconditioner subdomain the original cells, then its level-1 cells, then level 2, ...;
    `extras["waiwera_order"]` lists the new indices in the reference's cell order (original cells,
    then all level-1 cells in zone order, then level 2, ...) for output."""
    no = base.n_owned
    assert base.n_halo == 0
    sp = base.sub_ptr.astype(np.int64)
    nsub = sp.size - 1
    sub_of = np.repeat(np.arange(nsub), np.diff(sp))
    # matrix cells to create: (original cell, level, zone)
    entries = []
    for z, zone in enumerate(zones):
        for c in np.asarray(zone["cells"], dtype=np.int64):
            for lev in range(1, zone["geometry"].num_levels + 1):
                entries.append((int(sub_of[c]), lev, int(c), z))
    entries.sort(key=lambda e: (e[0], e[1]))          # stable: zone / cell order inside a level
    added = np.bincount([e[0] for e in entries], minlength=nsub)
    new_sp = np.concatenate([[0], np.cumsum(np.diff(sp) + added)])
    new_of_orig = new_sp[sub_of] + (np.arange(no) - sp[sub_of])
    n_new = int(new_sp[-1])
    pos = (new_sp[:-1] + np.diff(sp)).copy()           # next free slot per subdomain
    matrix_new = {}
    for s, lev, c, z in entries:
        matrix_new[(c, lev)] = int(pos[s])
        pos[s] += 1

    def remap(idx):
        idx = np.asarray(idx, dtype=np.int64)
        out = idx + (n_new - no)                       # boundary cells follow the owned ones
        own = idx < no
        out[own] = new_of_orig[idx[own]]
        return out
    m = LocalMesh(dims=base.dims, spacing=base.spacing, part=base.part, rank=base.rank, brick=base.brick,
                  n_global=base.n_global)
    m.n_owned, m.n_halo, m.n_bc = n_new, 0, base.n_bc
    cg, rock = np.zeros((m.n_local, 4)), np.zeros((m.n_local, 8))
    cg[remap(np.arange(base.n_local))] = base.cell_geom
    rock[remap(np.arange(base.n_local))] = base.rock
    level = np.zeros(n_new, dtype=np.int32)
    parent = np.full(n_new, -1, dtype=np.int64)
    parent[new_of_orig] = np.arange(no)
    fc, fg = [remap(base.face_cells.ravel()).reshape(-1, 2)], [base.face_geom]
    for zone in zones:
        geo = zone["geometry"]
        for c in np.asarray(zone["cells"], dtype=np.int64):
            v0 = base.cell_geom[c, 3]
            f = new_of_orig[c]
            cg[f, 3] = v0 * geo.volume[0]
            if zone.get("fracture_rock") is not None:
                rock[f] = zone["fracture_rock"]
            prev = f
            for lev in range(1, geo.num_levels + 1):
                q = matrix_new[(int(c), lev)]
                cg[q, 0:3] = base.cell_geom[c, 0:3]
                cg[q, 3] = v0 * geo.volume[lev]
                rock[q] = zone["matrix_rock"]
                level[q], parent[q] = lev, c
                g = np.zeros(12)
                g[0] = v0 * geo.connection_area[lev - 1]
                g[1], g[2] = geo.connection_distance[lev - 1], geo.connection_distance[lev]
                g[3] = g[1] + g[2]
                g[8:11] = base.cell_geom[c, 0:3]
                g[11] = 1.0
                fc.append(np.array([[prev, q]]))
                fg.append(g[None, :])
                prev = q
    m.cell_geom, m.rock = cg, rock
    # interior faces first, boundary faces last (the order the library expects)
    nint = base.n_faces - base.n_bc
    allc, allg = np.concatenate(fc), np.concatenate(fg)
    order = np.concatenate([np.arange(nint), np.arange(base.n_faces, allc.shape[0]), np.arange(nint, base.n_faces)])
    m.face_cells = allc[order].astype(np.int32)
    m.face_geom = allg[order]
    m.n_faces = m.face_cells.shape[0]
    m.sub_ptr = new_sp.astype(np.int32)
    m.bc_primary, m.bc_region = base.bc_primary, base.bc_region
    m.owned_gid = np.full(n_new, -1, dtype=np.int64)
    m.owned_gid[new_of_orig] = base.owned_gid if base.owned_gid is not None else np.arange(no)
    m.extras = dict(base.extras)
    m.extras["minc_level"], m.extras["minc_parent"] = level, parent
    m.extras["fracture_index"] = new_of_orig
    worder = list(new_of_orig)
    maxlev = max(z["geometry"].num_levels for z in zones)
    for lev in range(1, maxlev + 1):
        for zone in zones:
            if lev <= zone["geometry"].num_levels:
                worder += [matrix_new[(int(c), lev)] for c in np.asarray(zone["cells"], dtype=np.int64)]
    m.extras["waiwera_order"] = np.array(worder, dtype=np.int64)
    m.n_src = base.n_src
    if base.n_src:
        m.src_cell = new_of_orig[base.src_cell].astype(np.int32)
        m.src_rate, m.src_enthalpy, m.src_component = base.src_rate, base.src_enthalpy, base.src_component
    return m


def default_rock(n):
    """Reference default rock (src/rock.F90:69-76): k 1e-13, cond 2.5, phi 0.1, 2200, 1000."""
    r = np.zeros((n, 8))
    r[:, 0:3] = 1.0e-13
    r[:, 3:5] = 2.5
    r[:, 5] = 0.1
    r[:, 6] = 2200.0
    r[:, 7] = 1000.0
    return r


def heterogeneous_rock(n_global, seed=20250418):
    """SURVEY.md section 8d rock: k_x = k_y = 10^(-13 + 0.5 xi), k_z = 0.1 k_x, xi ~ N(0,1)."""
    xi = np.random.default_rng(seed).standard_normal(n_global)

    def fn(gid):
        r = default_rock(gid.size)
        kx = 10.0 ** (-13.0 + 0.5 * xi[gid])
        r[:, 0] = kx
        r[:, 1] = kx
        r[:, 2] = 0.1 * kx
        return r
    return fn


def liquid_density_estimate(t):
    """Rough liquid-water density (kg/m3) for building hydrostatic initial columns only."""
    return 1001.1 - 0.0867 * t - 0.0035 * t * t


def benchmark_initial_state(grid, ijk, eos="we", lens=True):
    """SURVEY.md section 8d initial state for natural cells ijk (n,3): returns primaries
    (unscaled, (n, np)) and regions.  P hydrostatic from 1e5 Pa at z = 0, T = 20 + 0.08*depth;
    optional two-phase lens (region 4, S_v 0.1..0.5) at depth 400..500 m, r < 200 m."""
    nx, ny, nz = grid.dims
    dx, dy, dz = grid.spacing
    depth_c = (np.arange(nz) + 0.5) * dz
    t_c = 20.0 + 0.08 * depth_c
    rho_c = liquid_density_estimate(t_c)
    p_c = np.empty(nz)
    p = 1.0e5
    for k in range(nz):
        p_c[k] = p + 0.5 * dz * GRAVITY * rho_c[k]
        p = p + dz * GRAVITY * rho_c[k]
    i, j, k = ijk[:, 0], ijk[:, 1], ijk[:, 2]
    P = p_c[k]
    T = t_c[k]
    region = np.ones(ijk.shape[0], dtype=np.int32)
    if eos == "w":
        return P[:, None].copy(), region
    second = T.copy()
    if eos in ("wce", "wae"):
        # CO2 partial pressure 2 % of the total pressure
        return np.stack([P, second, 0.02 * P], axis=1), region
    if eos in ("wse", "wsce", "wsae"):
        # salt family: 5 % salt everywhere, the two-phase lens as for eos we, and halite (region 5,
        # solid saturation 2 %) in a sprinkling of single-phase cells of the layer above the bottom
        third = np.full(ijk.shape[0], 0.05)
    if lens:
        x = (i + 0.5) * dx - 0.5 * nx * dx
        y = (j + 0.5) * dy - 0.5 * ny * dy
        r = np.sqrt(x * x + y * y)
        depth = depth_c[k]
        inl = (depth >= 400.0) & (depth <= 500.0) & (r < 200.0)
        region[inl] = 4
        second[inl] = 0.1 + 0.4 * r[inl] / 200.0
    if eos in ("wse", "wsce", "wsae"):
        hal = (region == 1) & (k == max(nz - 2, 0)) & ((i + 2 * j) % 5 == 0)
        region[hal] = 5
        third[hal] = 0.02
        if eos != "wse":     # gas partial pressure 2 % of the total pressure
            return np.stack([P, second, third, 0.02 * P], axis=1), region
        return np.stack([P, second, third], axis=1), region
    return np.stack([P, second], axis=1), region


def benchmark_sources(grid, co2_fraction=0.0):
    """4 injectors (10 kg/s, 1.0e6 J/kg) and 4 producers (-5 kg/s) at fixed box fractions."""
    nx, ny, nz = grid.dims
    out = []
    fr = [(0.25, 0.25), (0.75, 0.25), (0.25, 0.75), (0.75, 0.75)]
    for fx, fy in fr:
        out.append({"ijk": (int(fx * nx), int(fy * ny), int(0.7 * nz)), "rate": 10.0,
                    "enthalpy": 1.0e6, "component": 1})
        if co2_fraction > 0.0:  # injectors carry 5 % CO2 (component 2) in configs 4/5
            out[-1]["rate"] = 10.0 * (1.0 - co2_fraction)
            out.append({"ijk": out[-1]["ijk"], "rate": 10.0 * co2_fraction, "enthalpy": 1.0e6, "component": 2})
    fr2 = [(0.5, 0.25), (0.25, 0.5), (0.75, 0.5), (0.5, 0.75)]
    for fx, fy in fr2:
        out.append({"ijk": (int(fx * nx), int(fy * ny), int(0.4 * nz)), "rate": -5.0,
                    "enthalpy": 0.0, "component": 0})
    return out


def partition_shape(n):
    """Rank grid for n GPUs: 8 -> 2x2x2, 4 -> 2x2x1, 2 -> 2x1x1."""
    shapes = {1: (1, 1, 1), 2: (2, 1, 1), 4: (2, 2, 1), 8: (2, 2, 2), 16: (4, 2, 2)}
    if n in shapes:
        return shapes[n]
    p = [1, 1, 1]
    a = 0
    while n > 1:
        for q in (2, 3, 5, 7):
            if n % q == 0:
                p[a % 3] *= q
                n //= q
                a += 1
                break
        else:
            p[a % 3] *= n
            n = 1
    return tuple(p)


def row_mesh_1d(edges, thickness, radial=False, height=1.0, rock_record=None, inner_bc=None,
                outer_bc=None, sources=None, chunk=512):
    """One row of cells of a 2-D mesh as the reference's 1-D benchmark problems use it: cell i spans
    edges[i]..edges[i+1].  radial=True (`"mesh": {"radial": true}`): geometry by Pappus' theorem
    like src/mesh.F90:369-432, volume = dr * thickness * 2 pi r_c, face area = thickness * 2 pi r_f
    (`thickness` is then the vertical extent).  Cartesian (`"mesh": {"thickness": t}`, :355-364,
    404-414): volume = dx * height * t, face area = height * t.  Faces are vertical planes: the
    gravity term g.n is zero and the permeability direction is 1.  inner_bc / outer_bc = (primary,
    region) put a Dirichlet ghost cell on the first / last face (distances (d, 0),
    src/mesh.F90:647-651; boundary cells follow the owned cells in that order).  sources:
    [{cell, rate, enthalpy, component}].  Preconditioner subdomains: chunks of consecutive cells."""
    r = np.asarray(edges, dtype=np.float64)
    n = r.size - 1
    rc, dr = 0.5 * (r[1:] + r[:-1]), np.diff(r)

    def area(x):
        return thickness * 2.0 * np.pi * x if radial else height * thickness + 0.0 * x

    zc = -0.5 * (thickness if radial else height)
    m = LocalMesh(dims=(n, 1, 1), spacing=(float(dr[0]), 0.0, float(thickness)), part=(1, 1, 1), rank=0,
                  brick=(chunk, 1, 1), n_global=n)
    m.n_owned, m.n_halo = n, 0
    fc = np.stack([np.arange(n - 1), np.arange(1, n)], axis=1)
    fg = np.zeros((n - 1, 12))
    fg[:, 0] = area(r[1:-1])
    fg[:, 1] = 0.5 * dr[:-1]
    fg[:, 2] = 0.5 * dr[1:]
    fg[:, 3] = fg[:, 1] + fg[:, 2]
    fg[:, 4] = 1.0
    fg[:, 8] = r[1:-1]
    fg[:, 10] = zc
    fg[:, 11] = 1
    bcs = []
    for which, bc in (("inner", inner_bc), ("outer", outer_bc)):
        if bc is None:
            continue
        cell = 0 if which == "inner" else n - 1
        xf = r[0] if which == "inner" else r[-1]
        g = np.zeros((1, 12))
        g[0, 0] = area(np.array(xf))
        g[0, 1] = 0.5 * dr[cell]
        g[0, 3] = 0.5 * dr[cell]
        g[0, 4] = -1.0 if which == "inner" else 1.0
        g[0, 8] = xf
        g[0, 10] = zc
        g[0, 11] = 1
        fc = np.concatenate([fc, np.array([[cell, n + len(bcs)]])])
        fg = np.concatenate([fg, g])
        bcs.append((bc, xf))
    m.n_bc = len(bcs)
    if bcs:
        m.bc_primary = np.array([np.asarray(b[0][0], dtype=np.float64) for b in bcs])
        m.bc_region = np.array([int(b[0][1]) for b in bcs], dtype=np.int32)
    m.face_cells, m.face_geom, m.n_faces = fc.astype(np.int32), fg, fc.shape[0]
    cg = np.zeros((n + m.n_bc, 4))
    cg[:n, 0], cg[:n, 2] = rc, zc
    cg[:n, 3] = dr * thickness * 2.0 * np.pi * rc if radial else dr * height * thickness
    for k, b in enumerate(bcs):
        cg[n + k, 0], cg[n + k, 2] = b[1], zc
    m.cell_geom = cg
    rock = np.zeros((n + m.n_bc, 8))
    rock[:] = default_rock(1)[0] if rock_record is None else np.asarray(rock_record, dtype=np.float64)
    m.rock = rock
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


def radial_mesh_1d(r_edges, thickness, rock_record=None, outer_bc=None, sources=None, chunk=512):
    """1-D radial row (row_mesh_1d with radial=True)."""
    return row_mesh_1d(r_edges, thickness, radial=True, rock_record=rock_record, outer_bc=outer_bc,
                       sources=sources, chunk=chunk)


def column_mesh_1d(z_edges, area, rock=None, top_bc=None, sources=None, chunk=512, perm_direction=3):
    """A vertical column of cells, top first: cell i spans z_edges[i]..z_edges[i+1] (descending
    elevations) with horizontal cross-section `area`.  Face i joins cell i (above) to cell i+1
    (below), normal pointing down, so the gravity term g.n is +9.8; the Dirichlet ghost of top_bc =
    (primary, region) sits on the top face (normal up, g.n = -9.8, distances (d, 0)).
    perm_direction: rock permeability component across the faces (3 for a 3-D column, 2 for the
    y-vertical 2-D meshes of the reference's column benchmarks).  rock: one (8,) record or (n, 8)."""
    z = np.asarray(z_edges, dtype=np.float64)
    n = z.size - 1
    dz = -np.diff(z)
    zc = 0.5 * (z[1:] + z[:-1])
    m = LocalMesh(dims=(1, 1, n), spacing=(0.0, 0.0, float(dz[0])), part=(1, 1, 1), rank=0,
                  brick=(1, 1, chunk), n_global=n)
    m.n_owned, m.n_halo = n, 0
    fc = np.stack([np.arange(n - 1), np.arange(1, n)], axis=1)
    fg = np.zeros((n - 1, 12))
    fg[:, 0] = area
    fg[:, 1] = 0.5 * dz[:-1]
    fg[:, 2] = 0.5 * dz[1:]
    fg[:, 3] = fg[:, 1] + fg[:, 2]
    fg[:, 6] = -1.0
    fg[:, 7] = GRAVITY
    fg[:, 10] = z[1:-1]
    fg[:, 11] = perm_direction
    m.n_bc = 0
    if top_bc is not None:
        g = np.zeros((1, 12))
        g[0, 0] = area
        g[0, 1] = 0.5 * dz[0]
        g[0, 3] = 0.5 * dz[0]
        g[0, 6] = 1.0
        g[0, 7] = -GRAVITY
        g[0, 10] = z[0]
        g[0, 11] = perm_direction
        fc = np.concatenate([fc, np.array([[0, n]])])
        fg = np.concatenate([fg, g])
        prim, region = top_bc
        m.bc_primary = np.asarray(prim, dtype=np.float64)[None, :]
        m.bc_region = np.array([int(region)], dtype=np.int32)
        m.n_bc = 1
    m.face_cells, m.face_geom, m.n_faces = fc.astype(np.int32), fg, fc.shape[0]
    cg = np.zeros((n + m.n_bc, 4))
    cg[:n, 2] = zc
    cg[:n, 3] = area * dz
    if m.n_bc:
        cg[n, 2] = z[0]
    m.cell_geom = cg
    rk = np.zeros((n + m.n_bc, 8))
    rock = default_rock(1)[0] if rock is None else np.asarray(rock, dtype=np.float64)
    rk[:n] = rock
    if m.n_bc:
        rk[n] = rk[0]
    m.rock = rk
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
