"""stdin: tools/rocpd_summary.py CSV -> `kernel name (first 50 chars)  calls  avg us` per line (names hold commas: fields from the right)."""
import sys

for ln in sys.stdin.read().splitlines()[1:int(sys.argv[1]) if len(sys.argv) > 1 else 12]:
    f = ln.rsplit(",", 13)
    print("%-52s calls %6s  avg %8.2f us  vgpr %s" % (f[0].strip('"')[:50], f[1], float(f[3]) / 1e3, f[7]))
