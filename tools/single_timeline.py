#!/usr/bin/env python3
"""tools/single_timeline.py <results.db>: the dispatches of the LAST one-query calls in a rocprofv3 kernel trace, start / end relative to the
call's first kernel (where the time of a call goes besides its kernels: the gaps between dependent launches)."""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
s_col = "start" if "start" in cols else [c for c in cols if "start" in c][0]
e_col = "end" if "end" in cols else [c for c in cols if "end" in c][0]
rows = db.execute(f"select name, {s_col}, {e_col}, grid_x from kernels where name like '%pqv%' order by {s_col}").fetchall()
names = [re.sub(r"\(.*", "", r[0]).replace("void ", "")[:60] for r in rows]
# one-query calls: probe_single_kernel starts a call
starts = [i for i, n in enumerate(names) if "probe_single_kernel" in n]
for si in starts[-4:]:
    t0 = rows[si][1]
    print("call:")
    j = si
    prev_end = t0
    while j < len(rows) and (j == si or "probe_single_kernel" not in names[j]) and j < si + 8:
        n, s, e, g = names[j], rows[j][1], rows[j][2], rows[j][3]
        print(f"  {n:<62} start {(s - t0) / 1e3:8.2f}  end {(e - t0) / 1e3:8.2f}  dur {(e - s) / 1e3:7.2f}  gap before {(s - prev_end) / 1e3:6.2f}")
        prev_end = e
        j += 1
