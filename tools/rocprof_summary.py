"""Summarise a rocprofv3 rocpd database (kernel trace and/or PMC counters) into small text/JSON
files that fit the gpurun_out merge limit.  Usage: rocprof_summary.py <db> <out.txt> [--pmc]"""
import json
import re
import sqlite3
import sys


def short(name):
    return re.sub(r"\(.*", "", name)[:100]


def main():
    db, out = sys.argv[1], sys.argv[2]
    cur = sqlite3.connect(db).cursor()
    lines = []
    if "--pmc" in sys.argv:
        rows = list(cur.execute("select kernel_name, counter_name, count(*), avg(value), min(value), max(value) "
                                "from counters_collection group by kernel_name, counter_name order by 4 desc"))
        lines.append("kernel | counter | dispatches | avg | min | max")
        for n, c, k, a, mi, ma in rows[:40]:
            lines.append("%s | %s | %d | %.3f | %.3f | %.3f" % (short(n), c, k, a, mi, ma))
        js = {short(n) + "::" + c: {"dispatches": k, "avg": a} for n, c, k, a, mi, ma in rows}
        json.dump(js, open(out.replace(".txt", ".json"), "w"), indent=1)
    else:
        rows = list(cur.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3, min(end-start)/1e3, "
                                "max(end-start)/1e3 from kernels group by name order by 3 desc"))
        tot = sum(r[2] for r in rows) or 1.0
        lines.append("kernel | calls | total_ms | avg_us | min_us | max_us | pct")
        for n, c, t, a, mi, ma in rows[:30]:
            lines.append("%s | %d | %.2f | %.2f | %.2f | %.2f | %.1f%%" % (short(n), c, t, a, mi, ma, 100 * t / tot))
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:14]))


if __name__ == "__main__":
    main()
