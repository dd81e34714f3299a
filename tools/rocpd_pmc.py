# -*- coding: utf-8 -*-
"""Per-kernel sums of rocprofv3 --pmc counters from a rocpd SQLite database."""
import sqlite3
import sys


def table(db):
    c = sqlite3.connect(db)
    rows = c.execute(
        "select name, counter_name, count(*), avg(counter_value), avg(duration) from pmc_events "
        "group by name, counter_name order by name, counter_name").fetchall()
    return rows


if __name__ == "__main__":
    for db in sys.argv[1:]:
        for r in table(db):
            if "clr" in r[0]:
                # (name | counter | dispatches | average per dispatch | average duration; the name long enough to tell the
                #  template instantiations apart)
                print("%-120s| %-26s n=%-3d avg=%-16.6g avg_dur_us=%.1f" % (r[0][:120], r[1], r[2], r[3], r[4] / 1e3))
