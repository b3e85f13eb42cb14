"""Host-side helpers of bench.py that decide what the line says (no GPU): which committed PMC pass a launch's `traffic` is
taken from, the reference-default preconditioner block, the byte counts, the host description."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_traffic_is_taken_from_the_pass_of_the_launch_that_is_reported():
    """the composed second launch and the first launch have their own PMC figures; a profile of another mesh, brick shape or
    kernel gives None instead of a stale number"""
    b = _bench()
    first = b.traffic_from_profiles("c3", (216, 216, 216), (16, 16, 2), "k_pc_park<spmv,col16>", brick_order="tile4x4")
    second = b.traffic_from_profiles("c3", (216, 216, 216), (16, 16, 2), "k_pc_park<spmv,col16,composed>", composed=True, brick_order="tile4x4")
    assert first and second and second[0] > first[0] > 2.9e9
    assert "k_pc_park<true, true" in second[1] and "k_pc_park<true, false" in first[1] and "r6" in second[1]
    alg2 = b.pc_bytes(70263936, 10077696, 2) + 8 * 2 * 10077696
    assert 1.0 < second[0] / alg2 < 1.07                 # round 5's verdict bar for the composed launch
    # the bricks' numbering is part of the match: round 6's pass was made with 4 x 4 columns, round 5's x fastest
    x5 = b.traffic_from_profiles("c3", (216, 216, 216), (16, 16, 2), "k_pc_park<spmv,col16,composed>", composed=True, brick_order="x")
    assert x5 and "r5" in x5[1] and x5[0] > second[0]
    assert b.traffic_from_profiles("c3", (108, 108, 108), (16, 16, 2), "k_pc_park<spmv,col16>") is None
    assert b.traffic_from_profiles("c3", (216, 216, 216), (16, 16, 2), "k_pc_park<spmv,col16>", brick_order="z") is None
    assert b.traffic_from_profiles("c3", (216, 216, 216), (16, 16, 2), "k_pc_rows<2,spmv,3+3>") is None
    old = b.traffic_from_profiles("c3", (216, 216, 216), (16, 16, 2), "k_pc_park<spmv>")      # int32 column planes: other bytes,
    assert old and "r4" in old[1] and old[0] > first[0]                                            # so round 4's pass of THAT kernel
    w = b.traffic_from_profiles("c4", (172, 172, 170), (8, 4, 2), "k_pc_wave<3,spmv,composed>", composed=True, brick_order="tile4x4")
    assert w and "k_pc_wave<3, true, true" in w[1]


def test_reference_default_preconditioner_block():
    b = _bench()
    d = b.pc_reference_default("c3", "bjacobi")
    assert d["this_run"] == "bjacobi" and "asm" in d["reference_default"] and "timestepper.F90:2019-2020" in d["reference_default"]
    m = d["measured"]
    assert m["asm"]["krylov_iterations"] < m["bjacobi"]["krylov_iterations"]            # overlap saves iterations ...
    assert m["asm"]["ms_per_solve"] > 3.0 * m["bjacobi"]["ms_per_solve"]                # ... and costs more than it saves
    assert b.pc_reference_default("c5", "bjacobi")["measured"] is None                  # not measured there: says so
    raw = json.load(open(os.path.join(ROOT, "profiles", "pc_compare_r6.json")))
    assert raw["first_system"]["c3"] == m


def test_byte_counts_and_host_description():
    b = _bench()
    nnzb, n, bs = 70263936, 10077696, 2
    assert b.spmv_bytes(nnzb, n, bs) == nnzb * 36 + 4 * (n + 1) + 32 * n          # SURVEY section 8d's B_spmv
    assert b.pc_bytes(nnzb, n, bs) == nnzb * 36 + n * (4 + 48)
    p = b.physical_cores()
    assert p is None or (isinstance(p, int) and 1 <= p <= (os.cpu_count() or 1))
    # the bricks' numbering: "auto" keeps x fastest where the vertical neighbours are within 64 positions, else 4 x 4 columns
    assert b.DEFAULT_BRICK_ORDER == "auto"
    assert b.resolve_brick_order("auto", (216, 216, 216), (1, 1, 1), (16, 16, 2)) == "tile4x4"      # 14 x 14 bricks per layer
    assert b.resolve_brick_order("auto", (216, 216, 216), (2, 2, 2), (16, 16, 2)) == "x"            # 7 x 7 per rank
    assert b.resolve_brick_order("auto", (100, 100, 100), (1, 1, 1), (16, 16, 2)) == "x"
    assert b.resolve_brick_order("auto", (172, 172, 170), (1, 1, 1), (8, 4, 2)) == "tile4x4"
    assert b.resolve_brick_order("z", (216, 216, 216), (1, 1, 1), (16, 16, 2)) == "z"
