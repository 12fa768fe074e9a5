"""Per-kernel PMC averages from a rocprofv3 rocpd SQLite database."""
import sqlite3, sys
from collections import defaultdict
db = sys.argv[1]
con = sqlite3.connect(db); cur = con.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
pe = [t for t in tabs if t.startswith("rocpd_pmc_event")][0]
ip = [t for t in tabs if t.startswith("rocpd_info_pmc")][0]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
cols = lambda t: [r[1] for r in cur.execute("pragma table_info(%s)" % t)]
if "-v" in sys.argv:
    for t in (pe, ip, kd): print(t, cols(t))
q = ("select s.kernel_name, p.name, avg(e.value), count(*), avg(d.end-d.start) from %s e join %s p on e.pmc_id = p.id "
     "join %s d on e.event_id = d.event_id join %s s on d.kernel_id = s.id group by s.kernel_name, p.name" % (pe, ip, kd, ks))
res = defaultdict(dict)
for name, pmc, val, n, dur in cur.execute(q):
    res[name][pmc] = val; res[name]["_dur_us"] = dur / 1e3; res[name]["_n"] = n
for name, d in sorted(res.items(), key=lambda kv: -kv[1]["_dur_us"]):
    if len(sys.argv) > 2 and sys.argv[2] not in name and sys.argv[2] != "-v": continue
    print(name[:100]); print("   ", {k: (round(v, 1) if isinstance(v, float) else v) for k, v in d.items()})
