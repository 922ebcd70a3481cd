"""Per-kernel sums of one PMC counter from a rocprofv3 rocpd database (+ schema dump if the expected
tables are missing).  usage: rocpd_pmc.py <results.db> <COUNTER>"""
import re
import sqlite3
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return n[:90]


def main(path, counter):
    db = sqlite3.connect(path)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    try:
        cols = [c[1] for c in db.execute("pragma table_info('pmc_events')")]
        assert "counter_value" in cols
    except Exception as e:
        print("# no pmc_events:", e, tabs)
        return
    print("# %s per kernel (rocprofv3 units: KiB; gfx950 counts a wide coalesced read at half its bytes, see MI355X_MICROARCH.md HBM)" % counter)
    print("%-90s %8s %16s %16s %14s" % ("kernel", "launches", "sum_KiB", "per_launch_MiB", "avg_dur_us"))
    for n, c, sm, d in db.execute("select name, count(*), sum(counter_value), avg(duration) from pmc_events where counter_name = ? "
                                  "group by name order by sum(counter_value) desc", (counter,)):
        print("%-90s %8d %16.6g %16.2f %14.1f" % (short(n), c, sm, sm / c / 1024.0, (d or 0) / 1e3))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
