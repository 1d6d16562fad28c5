"""dev tool: kernel time grouped by (kernel, grid size) from a rocprofv3 rocpd database."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute('select name, grid_x, grid_y, grid_z, count(*), sum(duration), avg(duration) from kernels '
                  'group by name, grid_x, grid_y, grid_z order by sum(duration) desc limit 45').fetchall()
for name, gx, gy, gz, n, s, a in rows:
    print(f'{name[:44]:44s} grid=({gx},{gy},{gz}) calls={n:5d} total_ms={s / 1e6:8.3f} avg_us={a / 1e3:8.1f}')
