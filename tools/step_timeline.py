"""One resident step of bench.py from a rocprofv3 --kernel-trace database: every launch with start / end (ms from the
step's first kernel), duration, queue, and the idle gap on the critical queue before it.
  rocprofv3 --kernel-trace -d DIR -o NAME -- python3 bench.py --no-cpu-baseline --no-side ...
  python tools/step_timeline.py DIR/NAME_results.db [first-kernel-substring]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
first = sys.argv[2] if len(sys.argv) > 2 else "dio_lowcut_fft_kernel"
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [x for x in tabs if x.startswith("rocpd_kernel_dispatch")][0]
ks = [x for x in tabs if x.startswith("rocpd_info_kernel_symbol")][0]
rows = db.execute(f"select s.kernel_name, d.start, d.end, d.queue_id from {kd} d join {ks} s on d.kernel_id=s.id "
                  "order by d.start").fetchall()


def short(n):
    m = re.search(r"wm\d*L?\d*([a-z0-9_]+_kernel)", n)
    return m.group(1) if m else n[:32]


names = [short(r[0]) for r in rows]
starts = [i for i, n in enumerate(names) if n == first]
good = [(a, b) for a, b in zip(starts[:-1], starts[1:])
        if not any("codec" in n or "copyBuffer" in n or "pcm16" in n for n in names[a:b])]
a, b = good[len(good) // 2]
t0 = rows[a][1]
print("step %.3f ms, %d launches" % ((rows[b][1] - t0) / 1e6, b - a))
busy_until = t0
for r, n in zip(rows[a:b], names[a:b]):
    gap = (r[1] - busy_until) / 1e3
    print("%8.3f %8.3f %7.3f q%d %s%s" % ((r[1] - t0) / 1e6, (r[2] - t0) / 1e6, (r[2] - r[1]) / 1e6, r[3], n,
                                       "   <- idle %.0f us" % gap if gap > 2 else ""))
    busy_until = max(busy_until, r[2])
