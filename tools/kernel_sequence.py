"""Print the kernel sequence of the last complete frame of a rocprofv3 (rocpd sqlite) trace around every `copyBuffer`
(which launch of ours does the runtime's copy belong to?).   python tools/kernel_sequence.py RESULTS.db [pattern]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else "copyBuffer"
rows = c.execute("select name, start, duration from kernels order by start").fetchall()
names = [r[0].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:40] for r in rows]
idx = [i for i, n in enumerate(names) if pat in n]
print(len(rows), "kernels,", len(idx), pat)
seen = {}
for i in idx:
    key = (names[i - 1] if i else "-", names[i + 1] if i + 1 < len(names) else "-")
    seen[key] = seen.get(key, 0) + 1
for k, v in sorted(seen.items(), key=lambda kv: -kv[1])[:20]:
    print(f"{v:5d}  after {k[0]:42s} before {k[1]}")
if idx:
    last = idx[-1]
    lo = max(0, last - 16)
    cols = [d[1] for d in c.execute("pragma table_info(kernels)").fetchall()]
    gx = "grid_x" if "grid_x" in cols else None
    for i in range(lo, min(len(rows), last + 4)):
        extra = ""
        if gx:
            extra = str(c.execute(f"select grid_x, workgroup_x from kernels order by start limit 1 offset {i}").fetchone())
        print(f"   {names[i]:44s} {rows[i][2] / 1e3:8.2f} us  {extra}")
