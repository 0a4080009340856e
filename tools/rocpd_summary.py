"""Turn a rocprofv3 (rocpd sqlite) result into a small text summary for profiles/."""
import sqlite3
import sys


def main(db, out):
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), "
                          "max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), max(grid_x), max(workgroup_x) "
                          "from kernels group by name order by sum(duration) desc"))
    tot = sum(r[2] for r in rows) or 1
    with open(out, 'w') as fh:
        fh.write('# rocprofv3 --kernel-trace --stats summary (%s)\n\n' % db.split('/')[-1])
        fh.write('| kernel | calls | total ms | avg us | min us | max us | % | vgpr | agpr | sgpr | lds B | scratch B | grid | wg |\n|---|---|---|---|---|---|---|---|---|---|---|---|---|---|\n')
        for r in rows:
            fh.write('| %s | %d | %.3f | %.1f | %.1f | %.1f | %.2f | %s | %s | %s | %s | %s | %s | %s |\n' %
                     (r[0][:60], r[1], r[2] / 1e6, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / tot, r[6], r[7], r[8], r[9], r[10], r[11], r[12]))
        big = list(c.execute("select grid_x, count(*), avg(duration) from kernels where name='agx_step_kernel' group by grid_x"))
        fh.write('\nagx_step_kernel by grid size (threads): ' + ', '.join('%d threads x%d: avg %.3f ms' % (g, n, d / 1e6) for g, n, d in big) + '\n')
    print(open(out).read())


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
