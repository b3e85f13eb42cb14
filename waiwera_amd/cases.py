"""The synthetic structured workloads of SURVEY.md section 8d (bench.py at the BASELINE sizes, the tests at sizes the
oracle finishes in seconds): seeded heterogeneous rock, hydrostatic + geothermal initial state with the two-phase
lens, 4 injectors / 4 producers, top Dirichlet boundary, optional MINC level."""
import numpy as np

from waiwera_amd import mesh as M


def make_case(dims=(8, 8, 8), brick=(4, 4, 4), eos="we", lens=False, sources=True, part=(1, 1, 1),
              rank=0, hetero=True, top_bc=True, minc=False, spacing=None, brick_order="x", order="hyperplane",
              balanced_bricks=False):
    if spacing is None:
        # the two-phase lens sits 400..500 m deep: stretch shallow test boxes so that their bottom
        # layer falls inside it (otherwise lens=True silently means no lens)
        spacing = (10.0, 10.0, 500.0 / dims[2]) if lens and dims[2] * 10.0 < 450.0 else (10.0, 10.0, 10.0)
    g = M.StructuredGrid(dims, spacing=spacing, brick=brick, part=part, brick_order=brick_order, order=order,
                         balanced_bricks=balanced_bricks)
    srcs = M.benchmark_sources(g, co2_fraction=0.05 if eos in ("wce", "wse", "wae", "wsce", "wsae") else 0.0) if sources else None   # wse: 5 % salt
    bc = None
    if top_bc:
        bc = {"we": ([1.0e5, 20.0], 1), "w": ([1.0e5], 1), "wce": ([1.0e5, 20.0, 0.02e5], 1), "wae": ([1.0e5, 20.0, 0.02e5], 1),
              "wse": ([1.0e5, 20.0, 0.05], 1), "wsce": ([1.0e5, 20.0, 0.05, 0.02e5], 1),
              "wsae": ([1.0e5, 20.0, 0.05, 0.02e5], 1)}[eos]
    rock = M.heterogeneous_rock(g.n_global) if hetero else None
    mspec = None
    if minc:  # fracture fraction 0.1, one matrix level, 3 planes, 50 m
        mrock = M.default_rock(1)[0]
        mrock[0:3] = 1.0e-16
        mspec = dict(geometry=M.MincGeometry([0.1, 0.9], [50.0, 50.0, 50.0]), matrix_rock=mrock)
    lm = g.local_mesh(rank, rock_fn=rock, top_bc=bc, sources=srcs, minc=mspec)
    prim, region = M.benchmark_initial_state(g, lm.extras["prim_ijk"], eos=eos, lens=lens)
    return g, lm, prim, region


def scaled(prim, region, eos="we"):
    sc = np.ones((9, prim.shape[1]))
    for r in (1, 2, 4, 5, 6, 8):
        sc[r, 0] = 1.0e6
        if prim.shape[1] > 1:
            sc[r, 1] = 1.0e2 if r not in (4, 8) else 1.0
    out = prim / sc[region]
    if eos in ("wsce", "wsae"):
        out[:, 3] = prim[:, 3] / prim[:, 0]
    if eos in ("wce", "wae"):  # adaptive partial-pressure scaling Pg / P (src/eos_wge.F90:639-655)
        out[:, 2] = prim[:, 2] / prim[:, 0]
    return out
