import sqlite3, sys
con = sqlite3.connect(sys.argv[1]); cur = con.cursor()
pat = sys.argv[2] if len(sys.argv) > 2 else '%'
for r in cur.execute("select name, grid_x/workgroup_x, count(*), avg(duration)/1000 from kernels where name like ? group by name, grid_x order by 4 desc", (pat,)):
    print('%-50s grid=%5d calls=%4d avg=%8.1f us' % (r[0][:50], r[1], r[2], r[3]))
