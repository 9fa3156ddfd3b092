import sqlite3, sys, re
c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [x for x in cols if "name" in x][0]
rows = c.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
names = [re.sub(r"\(anonymous namespace\)::", "", r[0])[:60] for r in rows]
# print a window of 40 kernels from the middle
mid = len(names) // 2
for i in range(mid, mid + 40):
    gap = (rows[i][1] - rows[i-1][2]) / 1e3
    print("%8.1f us  gap %6.1f  %s" % ((rows[i][2] - rows[i][1]) / 1e3, gap, names[i]))
