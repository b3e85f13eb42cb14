"""profiles/pmc_traffic_r1.json from the two PMC summaries of tools/gpu_profile_r1.sh
(rocprof_pmc_fetch_r1.txt, rocprof_pmc_write_r1.txt: `kernel | counter | dispatches | avg | ...`,
KiB per dispatch).  HBM bytes = 2 * FETCH_SIZE + WRITE_SIZE (gfx950 correction of
MI355X_MICROARCH.md, calibrated on k_bcgs_p: 3 vectors read, 1 written)."""
import json
import sys

sys.path.insert(0, ".")
from bench import pc_bytes, spmv_bytes  # noqa: E402


def table(path):
    out = {}
    for ln in open(path):
        f = [x.strip() for x in ln.split("|")]
        if len(f) >= 4 and f[1] in ("FETCH_SIZE", "WRITE_SIZE"):
            out[f[0]] = float(f[3])
    return out


def main(src="gpurun_out", dst="profiles/pmc_traffic_r1.json", dims=(216, 216, 216), brick=(16, 16, 2)):
    fe, wr = table(src + "/rocprof_pmc_fetch_r1.txt"), table(src + "/rocprof_pmc_write_r1.txt")

    def hbm(k):
        return 2.0 * fe[k] * 1024.0 + wr[k] * 1024.0
    n = dims[0] * dims[1] * dims[2]
    nnzb = 7 * n - 2 * (dims[0] * dims[1] + dims[1] * dims[2] + dims[0] * dims[2])
    b_pc, b_spmv = pc_bytes(nnzb, n, 2), spmv_bytes(nnzb, n, 2)
    pc = [k for k in fe if k.startswith("void wai::k_pc_park<true")][0]
    sp = "void wai::k_spmv<2>"
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) -- python "
                     "bench.py --steps 1 --warmup 0 --no-cpu --spmv-reps 10 on 1 x MI355X, %dx%dx%d eos_we, bricks %dx%dx%d"
                     % (dims + brick),
           "correction": "HBM bytes = 2*FETCH_SIZE[KiB]*1024 + WRITE_SIZE[KiB]*1024 (gfx950: FETCH_SIZE tallies 128-B "
                         "requests at 64 B; calibrated on k_bcgs_p: 3 vectors read = 483.7 MB -> 2*%.0f KiB; 1 vector "
                         "written = 161.2 MB -> %.0f KiB)" % (fe["wai::k_bcgs_p"], wr["wai::k_bcgs_p"]),
           "dims": list(dims), "brick": list(brick),
           "k_pc_hbm_bytes_per_launch": hbm(pc), "k_pc_algorithmic_bytes": b_pc,
           "k_pc_traffic_over_algorithmic": hbm(pc) / b_pc,
           "k_spmv_hbm_bytes_per_launch": hbm(sp), "k_spmv_algorithmic_bytes": b_spmv,
           "k_spmv_traffic_over_algorithmic": hbm(sp) / b_spmv,
           "k_bcgs_p_hbm_bytes_per_launch": hbm("wai::k_bcgs_p"),
           "k_jacobian_hbm_bytes_per_launch": hbm("void wai::k_jacobian<1>"),
           "k_residual_hbm_bytes_per_launch": hbm("void wai::k_residual<1>"),
           "note": "k_pc = k_pc_park<spmv> (pivot-scaled rows, upper blocks parked in LDS)"}
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
