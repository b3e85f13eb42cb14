#!/usr/bin/env python
"""Register / LDS / scratch table of the HIP kernels of one source file, from hipcc's
-Rpass-analysis=kernel-resource-usage remarks (no GPU needed).

    python tools/kernel_resources.py waiwera_amd/csrc/kernels_linalg.hip [name-filter] [-- extra hipcc flags]
"""
import re
import subprocess
import sys


def main():
    args = sys.argv[1:]
    extra = []
    if "--" in args:
        i = args.index("--")
        args, extra = args[:i], args[i + 1:]
    src = args[0]
    filt = args[1] if len(args) > 1 else ""
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-c", src, "-o", "/dev/null",
           "-Rpass-analysis=kernel-resource-usage"] + extra
    out = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in out.splitlines():
        m = re.search(r"remark: (?:\s*)Function Name: (\S+)", line)
        if m:
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            cur = {"name": re.sub(r"\(.*", "", name).replace("void wai::", "")}
            rows.append(cur)
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+(?:\[[^\]]*\])?):\s+(\S+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = m.group(2)
    print("%-58s %5s %5s %7s %8s %6s %9s" % ("kernel", "VGPR", "AGPR", "scratch", "waves/SIMD", "LDS", "SGPRspill"))
    for r in rows:
        if filt and filt not in r["name"]:
            continue
        print("%-58s %5s %5s %7s %8s %6s %9s" % (r["name"][:58], r.get("VGPRs", "?"), r.get("AGPRs", "?"), r.get("ScratchSize [bytes/lane]", "?"),
                                             r.get("Occupancy [waves/SIMD]", "?"), r.get("LDS Size [bytes/block]", "?"), r.get("SGPRs Spill", "?")))


if __name__ == "__main__":
    main()
