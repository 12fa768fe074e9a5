"""Timeline of the last steps of a traced bench run from a rocprofv3 rocpd database (dev tool): every dispatch of the window as
(start ms, duration ms, queue, kernel), plus the gaps on each queue.   python scripts/rocpd_timeline.py t.db [window_ms]"""
import sqlite3
import sys


def main(db, window_ms=16.0):
    con = sqlite3.connect(db)
    cur = con.cursor()
    tables = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    disp = [t for t in tables if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tables if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in cur.execute("pragma table_info(%s)" % disp)]
    scols = [r[1] for r in cur.execute("pragma table_info(%s)" % sym)]
    name_col = "kernel_name" if "kernel_name" in scols else "display_name"
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else cols[0])
    rows = list(cur.execute("select d.start, d.end, d.%s, s.%s from %s d join %s s on d.kernel_id = s.id order by d.start" % (qcol, name_col, disp, sym)))
    em = [r for r in rows if "em_batch_kernel" in r[3]]
    if len(em) < 8:
        print("no EM launches"); return
    t_end = em[-3][0]                       # (a few launches before the end: steady state, not the flush)
    t0 = em[-8][0]
    last = {}
    for st, en, q, name in rows:
        if st < t0 or st > t_end:
            continue
        gap = (st - last[q]) / 1e6 if q in last else 0.0
        last[q] = en
        short = name.split("(")[0].replace("_ZN12_GLOBAL__N_1", "").replace("12_GLOBAL__N_1", "")[:48]
        if en - st > 20e3 or "em_batch" in name:
            print("%9.3f ms  +%7.3f  gap %6.3f  q%-3s %s" % ((st - t0) / 1e6, (en - st) / 1e6, gap, q, short))


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 16.0)
