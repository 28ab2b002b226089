import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select name, start, end, grid_x from kernels order by start"))
for i, (n, s, e, g) in enumerate(rows):
    if (e - s) > 400e3 and ("CatArray" in n or "elementwise" in n or "reduce_kernel" in n or "copy" in n.lower()):
        print("----", i, round((e - s) / 1e3, 1), "us grid", g, n[:90])
        for j in range(max(0, i - 3), min(len(rows), i + 4)):
            if j != i:
                print("      ", j - i, round((rows[j][2] - rows[j][1]) / 1e3, 1), rows[j][0][:100])
