"""profiles/pmc_traffic.json from a counter file written by tools/gpu_pmc4.sh (per-kernel averages of FETCH_SIZE and WRITE_SIZE,
separate rocprofv3 --pmc passes, kernel-trace only).

    python tools/pmc_traffic.py profiles/r04a_pmc_counters.txt

HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE reports half of the bytes of a wide
coalesced read (MI355X_MICROARCH.md, HBM section: 128-byte requests tallied at 64 bytes); the unit of both counters is KB.
The in-iteration instantiation of each pass (the one with the most dispatches) is the one recorded.
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def parse(path):
    """{section: {kernel: {counter: (avg, dispatches)}}}"""
    out, sec, kern = {}, None, None
    for line in open(path):
        m = re.match(r"=== (\S+) ===", line)
        if m:
            sec = out.setdefault(m.group(1), {})
            continue
        if sec is None or not line.strip() or line.startswith("--"):
            continue
        m = re.match(r"\s+(\S+)\s+avg/dispatch\s+([0-9.eE+-]+)\s+dispatches\s+(\d+)", line)
        if m and kern is not None:
            sec[kern][m.group(1)] = (float(m.group(2)), int(m.group(3)))
        elif not line.startswith(" "):
            kern = line.strip()
            sec.setdefault(kern, {})
    return out


def main(path):
    data = parse(path)
    rel = os.path.relpath(os.path.abspath(path), ROOT)
    res = {"_note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace only; tools/gpu_pmc4.sh) over `python bench.py "
                    "--steps 40 --warmup 5 --no-graph --no-other`; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: gfx950's FETCH_SIZE reports half "
                    "of a wide coalesced read (MI355X_MICROARCH.md, HBM section); raw per-kernel averages in " + rel}
    for wl in ("dimp50", "prdimp50"):
        f, w = data.get(wl + "_fetch", {}), data.get(wl + "_write", {})
        rec = {}
        for short in ("k_corr2", "k_adj2"):
            cands = [(k, v["FETCH_SIZE"]) for k, v in f.items() if short in k and "FETCH_SIZE" in v]
            if not cands:
                continue
            name, (fetch, calls) = max(cands, key=lambda kv: kv[1][1])
            wr = w.get(name, {}).get("WRITE_SIZE")
            if wr is None:
                continue
            rec[short] = {"hbm_bytes_per_launch": int(round((2 * fetch + wr[0]) * 1024)),
                          "source": f"{rel}: {name[:48]} ({calls} dispatches) FETCH_SIZE {fetch:.1f} KB x2 + WRITE_SIZE {wr[0]:.1f} KB"}
        if rec:
            res[wl] = rec
    with open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w") as fh:
        json.dump(res, fh, indent=1)
        fh.write("\n")
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])
