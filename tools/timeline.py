#!/usr/bin/env python3
"""Kernel timeline of ONE training step from a rocprofv3 --kernel-trace database (rocpd sqlite):
    tools/timeline.py gpurun_out/<dir>/<name>_results.db [steps back from the end, default 3]
Prints start (us from the step's compositing backward), duration, the gap to the previous kernel's end, queue and name."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = list(db.execute(f"select k.kernel_name, d.start, d.end, d.queue_id from {disp} d join {sym} k on d.kernel_id = k.id order by d.start"))
idx = [i for i, r in enumerate(rows) if "render_bwd" in r[0]]
a, b = idx[-back - 1], idx[-back]
t0, prev_end, idle = rows[a][1], None, 0.0
for r in rows[a:b]:
    gap = (r[1] - prev_end) / 1e3 if prev_end else 0.0
    idle += max(gap, 0.0)
    print("%9.1f %8.1f gap %7.1f q%s %s" % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, gap, r[3], r[0][:90]))
    prev_end = max(prev_end or 0, r[2])
print("step %.1f us, idle %.1f us" % ((rows[b][1] - t0) / 1e3, idle + max(0.0, (rows[b][1] - prev_end) / 1e3)))
