"""Rewrite the `file:line` numbers of DESIGN.md section 4 from the current `__global__` declarations (run after editing kernels; checked by
tests/test_docs_cpu.py).  A cell `file.hip:a, b, c` is matched to the kernels named in its row that are declared in that file, in row order."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_docs_cpu import _kernel_lines  # noqa: E402


def main():
    path = os.path.join(ROOT, "DESIGN.md")
    text = open(path).read()
    a, b = text.index("## 4. Kernel inventory"), text.index("## 5. Launch graphs")
    decl = _kernel_lines()
    rows = text[a:b].split("\n")
    for r, row in enumerate(rows):
        if not row.startswith("| `k_"):
            continue
        cells = row.split("|")
        names = re.findall(r"`(k_\w+)", cells[1])

        def repl(m):
            fn, nums = m.group(1), m.group(2)
            if "-" in nums:
                lines = sorted(ln for k in names for f, ln in decl.get(k, []) if f == fn)
                return f"{fn}:{lines[0]}-{lines[-1]}" if lines else m.group(0)
            lines = [ln for k in names for f, ln in decl.get(k, []) if f == fn]
            want = len(re.findall(r"\d+", nums))
            return f"{fn}:{', '.join(str(x) for x in lines[:want])}" if len(lines) >= want else m.group(0)
        cells[2] = re.sub(r"(\w+\.(?:hip|h)):([0-9,\s\-]+[0-9])", repl, cells[2])
        rows[r] = "|".join(cells)
    open(path, "w").write(text[:a] + "\n".join(rows) + text[b:])


if __name__ == "__main__":
    main()
