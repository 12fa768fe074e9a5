"""Build profiles/rNN_pmc_traffic.json from the FETCH_SIZE / WRITE_SIZE passes of profile_round.sh.

    python3 scripts/make_traffic_json.py <prof dir> <out json>

FETCH_SIZE / WRITE_SIZE are KB per dispatch.  Reads are doubled as MI355X_MICROARCH.md prescribes for
gfx950 (128-byte requests of wide coalesced streams are counted as 64 B); WRITE_SIZE is reported as is."""
import json
import os
import sqlite3
import sys


def per_kernel(db):
    con = sqlite3.connect(db)
    cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    pe = [t for t in tabs if t.startswith("rocpd_pmc_event")][0]
    ip = [t for t in tabs if t.startswith("rocpd_info_pmc")][0]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    q = ("select s.kernel_name, p.name, avg(v), avg(dur) from (select e.event_id as ev, e.pmc_id as pm, sum(e.value) as v "
         "from %s e group by e.event_id, e.pmc_id) x join %s p on x.pm = p.id join (select event_id, kernel_id, "
         "end - start as dur from %s) d on x.ev = d.event_id join %s s on d.kernel_id = s.id group by s.kernel_name, p.name"
         % (pe, ip, kd, ks))
    out = {}
    for name, pmc, val, dur in cur.execute(q):
        out.setdefault(name, {})[pmc] = val
        out[name]["_ms"] = dur / 1e6
    return out


def first_db(d):
    for root, _, files in os.walk(d):
        for f in files:
            if f.endswith(".db"):
                return os.path.join(root, f)
    raise SystemExit("no rocpd database under " + d)


def main(prof, out):
    res = {"note": "rocprofv3 --pmc passes (scripts/profile_round.sh); FETCH_SIZE/WRITE_SIZE in KB per dispatch; "
                   "hbm_read_bytes = 2 x FETCH_SIZE (MI355X_MICROARCH.md: gfx950 counts the 128-B requests of wide "
                   "coalesced reads as 64 B); WRITE_SIZE as reported"}
    for key, tag in (("yud_102", "yud"), ("stress_512x1000x8x50", "stress")):
        f = per_kernel(first_db(os.path.join(prof, tag + "_fetch")))
        w = per_kernel(first_db(os.path.join(prof, tag + "_write")))
        for kname, short in (("em_batch_kernel", ""), ("conv_gemm_dma_kernelILi2ELi2ELi2ELi2ELb0", "_conv"),
                             ("conv5x5_winograd_kernel", "_conv2w"), ("conv3x3_winograd_kernel", "_conv3w"),
                             ("conv_pieces_kernelILi5E", "_conv2"), ("conv1_pieces_kernel", "_conv1")):
            fk = [k for k in f if kname in k]
            wk = [k for k in w if kname in k]
            if not fk or not wk:
                continue
            fs, ws = f[fk[0]].get("FETCH_SIZE", 0.0), w[wk[0]].get("WRITE_SIZE", 0.0)
            res[key + short] = {"kernel": {"": kname, "_conv": "conv_gemm_dma_kernel<2,2,2,2,false> (avg of its launches)",
                                           "_conv2w": "conv5x5_winograd_kernel (conv2)",
                                           "_conv3w": "conv3x3_winograd_kernel (avg of conv3/4/5)",
                                           "_conv2": "conv_pieces_kernel<5>(conv2)", "_conv1": "conv1_pieces_kernel"}[short],
                                "FETCH_SIZE_KB": fs, "WRITE_SIZE_KB": ws, "hbm_read_bytes": 2e3 * fs,
                                "hbm_write_bytes": 1e3 * ws, "kernel_ms_profiled": f[fk[0]]["_ms"]}
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
