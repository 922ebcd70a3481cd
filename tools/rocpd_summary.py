"""Summarise a rocprofv3 rocpd SQLite result (ROCm 7.2 default output of `--kernel-trace --stats`)
into the text table committed under profiles/.  usage: rocpd_summary.py <results.db> [> out.txt]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name if len(name) <= 110 else name[:107] + "..."


def main(path):
    db = sqlite3.connect(path)
    rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), max(sgpr_count), "
                      "max(lds_size), max(scratch_size) from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    print("# rocprofv3 --kernel-trace --stats summary of %s" % path)
    print("%-110s %7s %12s %12s %12s %12s %6s %5s %5s %7s %7s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct", "vgpr", "sgpr", "lds", "scratch"))
    for n, c, s, a, mn, mx, v, sg, lds, scr in rows:
        print("%-110s %7d %12.1f %12.1f %12.1f %12.1f %6.2f %5s %5s %7s %7s" % (short(n), c, s / 1e3, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * s / tot, v, sg, lds, scr))
    try:
        pm = db.execute("select count(*) from pmc_events").fetchone()[0]
        if pm:
            print("\n# PMC counters (sum over dispatches per kernel)")
            for r in db.execute("select k.name, p.counter_name, count(*), sum(p.value) from pmc_events p join kernels k on p.event_id = k.id "
                                "group by k.name, p.counter_name order by k.name").fetchall():
                print("%-90s %-24s n=%-6d sum=%.6g" % (short(r[0])[:90], r[1], r[2], r[3]))
    except Exception as e:  # schema differences between ROCm versions
        print("# (no PMC table: %s)" % e)


if __name__ == "__main__":
    main(sys.argv[1])
