"""pmc_traffic_<tag>.json from the two PMC summaries of tools/gpu_profile.sh
(rocprof_pmc_fetch_<tag>.txt, rocprof_pmc_write_<tag>.txt: `kernel | counter | dispatches | avg | ...`,
KiB per dispatch) and the bench line of the same run (algorithmic bytes).  HBM bytes =
2 * FETCH_SIZE + WRITE_SIZE (gfx950 correction of MI355X_MICROARCH.md, calibrated on k_bcgs_p:
3 vectors read, 1 written).

    python tools/pmc_traffic.py <dir> <tag> <out.json>
"""
import json
import re
import sys


def table(path):
    out = {}
    for ln in open(path):
        f = [x.strip() for x in ln.split("|")]
        if len(f) >= 4 and f[1] in ("FETCH_SIZE", "WRITE_SIZE"):
            out[f[0]] = float(f[3])
    return out


def main(src, tag, dst):
    fe, wr = table("%s/rocprof_pmc_fetch_%s.txt" % (src, tag)), table("%s/rocprof_pmc_write_%s.txt" % (src, tag))
    bench = json.load(open("%s/bench_%s.json" % (src, tag)))
    roof = bench["roofline"]

    def hbm(k):
        return 2.0 * fe[k] * 1024.0 + wr[k] * 1024.0

    def find(pat):
        ks = [k for k in fe if re.search(pat, k) and k in wr]
        return max(ks, key=lambda k: fe[k]) if ks else None
    m = re.search(r"(\d+)x(\d+)x(\d+) structured", bench["config"]["workload"])
    dims = [int(v) for v in m.groups()]
    mb = re.search(r"\((\d+)x(\d+)x(\d+) bricks\)", bench["config"]["workload"])
    brick = [int(v) for v in mb.groups()]
    # the launch bench.py's roofline times: the fused operator on a stored operand (first half of an iteration); the
    # composed second launch of the 2 x 2 kernel (operand R - alpha V formed inside: one more vector read) beside it
    pc = find(r"k_pc_park<true, false|k_pc_rows<\d, true, \d, \d, false|k_pc_wave<\d, true, false|k_pc<\d, true") or \
        find(r"k_pc_park<true|k_pc_rows<\d, true|k_pc_wave<\d, true|k_pc<\d, true")
    pcx = find(r"k_pc_park<true, true|k_pc_wave<\d, true, true|k_pc_rows<\d, true, \d, \d, true")
    sp = find(r"k_spmv<")
    # since round 6 the line's roofline is the iteration's slower fused launch; the other one's figures travel as <half>_half_*
    dom = roof.get("dominant_half", "first")
    alg_first = roof["algorithmic_bytes_per_launch"] if dom == "first" else roof["first_half_algorithmic_bytes_per_launch"]
    alg_second = roof["algorithmic_bytes_per_launch"] if dom == "second" else roof.get("second_half_algorithmic_bytes_per_launch")
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) -- python "
                     "bench.py --config ... --lead 1 --steps 1 --warmup 0 --no-cpu --spmv-reps 10 on 1 x MI355X; " +
                     bench["config"]["workload"],
           "correction": "HBM bytes = 2*FETCH_SIZE[KiB]*1024 + WRITE_SIZE[KiB]*1024 (gfx950: FETCH_SIZE tallies 128-B "
                         "requests at 64 B; calibrated on k_bcgs_p: 3 vectors read -> 2*%.0f KiB; 1 vector written -> "
                         "%.0f KiB)" % (fe.get("wai::k_bcgs_p", 0.0), wr.get("wai::k_bcgs_p", 0.0)),
           "dims": dims, "brick": brick, "brick_order": bench["config"].get("brick_order", "x"), "k_pc_kernel": pc,
           "k_pc_hbm_bytes_per_launch": hbm(pc), "k_pc_algorithmic_bytes": alg_first,
           "k_pc_traffic_over_algorithmic": hbm(pc) / alg_first,
           "k_spmv_hbm_bytes_per_launch": hbm(sp), "k_spmv_algorithmic_bytes": roof["spmv_algorithmic_bytes_per_launch"],
           "k_spmv_traffic_over_algorithmic": hbm(sp) / roof["spmv_algorithmic_bytes_per_launch"]}
    if pcx and pcx != pc:
        m2 = re.search(r"\((\d+) cells\)", bench["config"]["workload"])
        ncell = int(m2.group(1)) if m2 else 0
        bs = 2 if "k_pc_park" in pcx else 3
        alg = alg_second or (alg_first + 8 * bs * ncell)
        out.update({"k_pc_composed_kernel": pcx, "k_pc_composed_hbm_bytes_per_launch": hbm(pcx),
                    "k_pc_composed_algorithmic_bytes": alg, "k_pc_composed_traffic_over_algorithmic": hbm(pcx) / alg})
    for name, pat in (("k_bcgs_p", r"k_bcgs_p"), ("k_jacobian", r"k_jacobian"), ("k_residual", r"k_residual"),
                      ("k_eos_pert", r"k_eos_pert")):
        k = find(pat)
        if k:
            out[name + "_hbm_bytes_per_launch"] = hbm(k)
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:4])
