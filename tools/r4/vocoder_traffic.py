#!/usr/bin/env python
"""HBM-side bytes of ONE vocoder decode from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over tools/r4/vocos_only.py: every
dispatch of the process is the vocoder's, so bytes per decode = sum over dispatches / decodes.  FETCH_SIZE is doubled (gfx950: wide reads
are tallied at half their bytes, MI355X_MICROARCH.md); both counters are in KB.

    python tools/r4/vocoder_traffic.py <fetch results.db> <write results.db> <decodes>  ->  JSON fragment"""
import json
import sqlite3
import sys


def total_kb(path, counter):
    db = sqlite3.connect(path)
    (v,) = db.execute("select sum(value) from counters_collection where counter_name = ? and kernel_name not like '%dft_basis%'", (counter,)).fetchone()
    (n,) = db.execute("select count(distinct dispatch_id) from counters_collection where counter_name = ?", (counter,)).fetchone()
    return float(v or 0.0), int(n)


f_kb, nf = total_kb(sys.argv[1], "FETCH_SIZE")
w_kb, nw = total_kb(sys.argv[2], "WRITE_SIZE")
dec = int(sys.argv[3])
print(json.dumps({"vocoder": {"fetch_kb_per_decode": round(f_kb / dec, 1), "write_kb_per_decode": round(w_kb / dec, 1), "dispatches_per_decode": round(nf / dec, 1),
                              "hbm_bytes_per_decode": int((2 * f_kb + w_kb) * 1024 / dec), "decodes": dec,
                              "_source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over tools/r4/vocos_only.py (L = 938); FETCH_SIZE doubled per the gfx950 correction"}}))
