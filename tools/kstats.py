"""Per-kernel summary of a rocprofv3 rocpd database (`rocprofv3 --kernel-trace -d DIR -o NAME`)."""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = c.execute('select name, count(*), avg(end-start), sum(end-start), min(end-start) from kernels group by name order by 4 desc').fetchall()
tot = sum(r[3] for r in rows)
print('%-70s %7s %10s %10s %6s' % ('kernel', 'calls', 'avg us', 'min us', '%'))
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 25]:
    print('%-70s %7d %10.1f %10.1f %6.1f' % (r[0][:70], r[1], r[2] / 1e3, r[4] / 1e3, 100 * r[3] / tot))
print('kernel time per step: %.3f ms' % (tot / 1e6 / steps))
