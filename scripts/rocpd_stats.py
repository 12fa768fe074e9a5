"""Summarise a rocprofv3 rocpd SQLite database into a per-kernel stats CSV (name, calls,
total/avg/min/max duration in microseconds, percentage) -- the same content as
`rocprofv3 --stats` kernel_stats.csv."""
import csv
import sqlite3
import sys


def main(db, out, skip_passes=0, passes=0):
    """skip_passes / passes: the traced program ran `passes` identical passes; drop every kernel's dispatches of
    the first `skip_passes` of them (warm-up: cold instruction caches, first-touch allocations)."""
    con = sqlite3.connect(db)
    cur = con.cursor()
    tables = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    disp = [t for t in tables if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tables if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in cur.execute("pragma table_info(%s)" % disp)]
    scols = [r[1] for r in cur.execute("pragma table_info(%s)" % sym)]
    name_col = "kernel_name" if "kernel_name" in scols else "display_name"
    if passes > 0 and skip_passes > 0:
        per = {}
        for name, start, dur in cur.execute("select s.%s, d.start, d.end - d.start from %s d join %s s on d.kernel_id = s.id "
                                            "order by d.start" % (name_col, disp, sym)):
            per.setdefault(name, []).append((start, dur))
        # What ran before the repeated passes -- the model load: weight re-packing and the calibration forwards of vpk_cnn_load, which
        # use kernels of the passes too (round 6: each calibration forward ends with absmax_kernel) -- is cut off first: everything up
        # to the last dispatch of a load-only kernel.  After that every kernel's dispatch count is a multiple of `passes`, and the
        # kept part starts with the first kept dispatch of any of them.  (Round 5's rule -- counts over the WHOLE trace -- found no
        # kernel whose count was a multiple under VPK_ALGORITHM=0 and wrote an empty summary: ADVICE r5.)
        load_only = ("pack_weights_kernel", "dense_tile_weights_kernel", "absmax_kernel")
        t_load = max([st for name, v in per.items() if any(k in name for k in load_only) for (st, _) in v] or [-1])
        starts = []
        for name, v in per.items():
            after = [st for (st, _) in v if st > t_load]
            if after and len(after) % passes == 0:
                starts.append(after[len(after) // passes * skip_passes])
        if not starts:
            raise SystemExit("rocpd_stats: no kernel with a dispatch count that is a multiple of %d after the model load -- "
                             "wrong --passes, or the trace is not one of repeated passes" % passes)
        t_keep = min(starts)
        rows = []
        for name, v in per.items():
            durs = [d for (st, d) in v if st >= t_keep]
            if not durs:
                continue
            rows.append((name, len(durs), sum(durs), sum(durs) / len(durs), min(durs), max(durs)))
        rows.sort(key=lambda r: -r[2])
    else:
        q = ("select s.%s, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start), "
             "max(d.end - d.start) from %s d join %s s on d.kernel_id = s.id group by s.%s order by 3 desc"
             % (name_col, disp, sym, name_col))
        rows = list(cur.execute(q))
    total = sum(r[2] for r in rows) or 1
    with open(out, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "MinUs", "MaxUs", "Percentage"])
        for r in rows:
            w.writerow([r[0], r[1], "%.1f" % (r[2] / 1e3), "%.2f" % (r[3] / 1e3), "%.2f" % (r[4] / 1e3),
                        "%.2f" % (r[5] / 1e3), "%.2f" % (100.0 * r[2] / total)])
    for r in rows[:14]:
        print("%-90s calls %4d avg %10.1f us  %5.1f%%" % (r[0][:90], r[1], r[3] / 1e3, 100.0 * r[2] / total))


if __name__ == "__main__":
    a = sys.argv
    main(a[1], a[2], int(a[a.index("--skip-passes") + 1]) if "--skip-passes" in a else 0,
         int(a[a.index("--passes") + 1]) if "--passes" in a else 0)
