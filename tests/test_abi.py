"""The C-ABI library loads without a GPU and exports every symbol include/*.h declares (waiwera_hip.h: the drop-in
boundary; waiwera_hip_bench.h: measurement and test entry points)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(headers=("waiwera_hip.h", "waiwera_hip_bench.h")):
    syms = set()
    for h in headers:
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        syms |= set(re.findall(r"\b(wai_[a-z_0-9]+)\s*\(", text))
    return sorted(syms)


def test_library_exports_all_declared_symbols():
    from waiwera_amd import build
    so = build.build()
    lib = ctypes.CDLL(so)
    syms = declared_symbols()
    assert len(syms) >= 40
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing


def test_python_binding_covers_header():
    from waiwera_amd import lib
    assert set(declared_symbols()) == set(lib.EXPORTED)


def test_fortran_module_binds_every_product_entry_point():
    """north_star's host is Fortran 2003 over iso_c_binding: whatever the product header declares has a bind(c) interface
    in waiwera_amd/fortran/waiwera_hip_module.F90 (the amdflang host of tests/test_hip_fortran.py compiles it)."""
    text = open(os.path.join(ROOT, "waiwera_amd", "fortran", "waiwera_hip_module.F90")).read()
    bound = set(re.findall(r'bind\(c,\s*name\s*=\s*"(wai_[a-z_0-9]+)"\)', text))
    missing = sorted(set(declared_symbols(("waiwera_hip.h",))) - bound)
    assert not missing, missing
    public = set(re.findall(r"\b(wai_[a-z_0-9]+)\b", " ".join(ln for ln in re.sub(r"&\s*\n", " ", text).splitlines()
                                                               if ln.strip().startswith("public ::"))))
    private = {"wai_ctx_create", "wai_ctx_destroy"}   # behind hip_flow_simulation_type%init / %destroy
    hidden = sorted(b for b in bound - public - private if not re.search(r"procedure.*hip_sim", b))
    # an interface that is neither public nor wrapped by a type-bound procedure is unreachable for a host
    wrapped = set(re.findall(r"=\s*(wai_[a-z_0-9]+)\(", text)) | set(re.findall(r"call\s+(wai_[a-z_0-9]+)\(", text))
    assert not [h for h in hidden if h not in wrapped], [h for h in hidden if h not in wrapped]


def test_headers_are_plain_c_and_a_c_host_links(tmp_path):
    """The boundary is a C ABI: both headers compile as C99 with warnings as errors, every declared function can be
    named from C, and a C host links against the library and runs its device-free entry points."""
    import subprocess
    from waiwera_amd import build
    so = build.build()
    names = declared_symbols()
    src = tmp_path / "host.c"
    src.write_text(
        '#include "waiwera_hip.h"\n#include "waiwera_hip_bench.h"\n#include <stdio.h>\n'
        "typedef void (*fn)(void);\n"
        "static const fn table[] = {" + ", ".join("(fn)%s" % n for n in names) + "};\n"
        "int main(void) {\n"
        "  wai_solver_opts o; wai_eos_desc e;\n"
        "  wai_default_opts(&o); wai_default_eos(&e, WAI_EOS_WE);\n"
        '  printf("%d %d %d %g\\n", (int)(sizeof table / sizeof table[0]), o.ksp_type, o.pc_type, o.ksp_rtol);\n'
        "  return table[0] == 0;\n}\n")
    exe = tmp_path / "host"
    libdir = os.path.dirname(so)
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                           str(src), "-o", str(exe), "-L", libdir, "-lwaiwera_hip", "-Wl,-rpath," + libdir,
                           "-Wl,-rpath,/opt/rocm/lib", "-Wl,--allow-shlib-undefined"])
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()
    assert int(out[0]) == len(names)
    assert float(out[3]) == 1e-5     # KSP rtol: PETSc's default, which the reference leaves alone (timestepper.F90:1677-1699)


def test_bench_entry_points_are_not_in_the_product_header():
    product = set(declared_symbols(("waiwera_hip.h",)))
    assert not [s for s in product if re.match(r"wai_(bench_|profile_|timer_|launch_stats|comm_stats)", s)], product


def test_defaults_match_reference_defaults():
    """wai_default_opts / wai_default_eos need no GPU: reference defaults
    (src/timestepper.F90:1567-1573,1998-2020; src/eos_we.F90:75-76)."""
    from waiwera_amd import lib
    o = lib.default_opts()
    assert (o.ksp_type, o.max_newton_its) == (0, 8)
    # the ONE default that is not the reference's, stated in the header: brick block Jacobi (the fused path) where Waiwera
    # defaults to PCASM overlap 1 (src/timestepper.F90:2019-2020); overlap and fill levels are PETSc's defaults
    assert (o.pc_type, o.asm_overlap, o.ilu_levels) == (lib.PC["bjacobi"], 1, 0)
    hdr = open(os.path.join(ROOT, "include", "waiwera_hip.h")).read()
    assert "NOT the reference's default" in hdr and "timestepper.F90:2019-2020" in hdr
    assert (o.ftol_rel, o.ftol_abs, o.utol_rel, o.utol_abs) == (1e-5, 1.0, 1e-10, 1.0)
    assert (o.fd_eps, o.fd_umin, o.ksp_rtol) == (1e-8, 1e-2, 1e-5)
    e = lib.eos_desc("we")
    assert (e.pressure_scale, e.temperature_scale, e.rp_type, e.cp_type) == (1e6, 1e2, 1, 0)


def test_no_product_code_touches_the_oracle():
    """The product path must not import, link or execute anything under oracle/."""
    bad = []
    for base in ("waiwera_amd", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".hip", ".cpp", ".hpp", ".h", ".F90")):
                    t = open(os.path.join(dp, f), errors="ignore").read()
                    if re.search(r"oracle_lib|liboracle|wai_oracle|\bwo_[a-z]+\(", t):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_assembly_kernels_and_oracle_round_alike():
    """The assembly / EOS kernels and the oracle are both built without FMA contraction: the FD Jacobian
    then does not depend on how a kernel orders its loops, and the device follows the oracle's Krylov
    counts."""
    from waiwera_amd import build as B
    assert "-ffp-contract=off" in B.PER_FILE.get("kernels_assembly.hip", [])
    mk = open(os.path.join(ROOT, "oracle", "Makefile")).read()
    assert "-ffp-contract=off" in mk
