"""Timeline of ONE frame of tools/frame_phases.py from a `rocprofv3 --hip-trace --kernel-trace --output-format csv` run: every HIP API call
of the host thread and every kernel, offsets in us from the frame's first API call.   python tools/frame_timeline.py DIR [frame index from the end]"""
import csv
import glob
import os
import sys


def load(pattern):
    f = glob.glob(os.path.join(sys.argv[1], "**", pattern), recursive=True)
    if not f:
        return []
    with open(f[0], newline="") as fh:
        return list(csv.DictReader(fh))


def main():
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    api = load("*hip_api_trace.csv")
    ker = load("*kernel_trace.csv")
    ker.sort(key=lambda r: int(r["Start_Timestamp"]))
    api.sort(key=lambda r: int(r["Start_Timestamp"]))
    ends = [int(r["End_Timestamp"]) for r in ker if "k_iou_final" in r["Kernel_Name"]]
    if len(ends) < back + 2:
        print("not enough frames in the trace", len(ends))
        return
    t_lo, t_hi = ends[-back - 2], ends[-back - 1]          # from the end of frame f-1 to the end of frame f
    rows = []
    for r in api:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if t_lo - 20000 <= s <= t_hi + 20000:
            rows.append((s, e, "API ", r["Function"]))
    for r in ker:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if t_lo - 20000 <= s <= t_hi + 20000:
            rows.append((s, e, "KERN", r["Kernel_Name"].split("(")[0][-40:]))
    rows.sort()
    print(f"frame window {(t_hi - t_lo) / 1e3:.1f} us (end of k_iou_final to end of the next k_iou_final); offsets from the previous k_iou_final's end")
    for s, e, kind, name in rows:
        print(f"{(s - t_lo) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  {kind} {name}")


if __name__ == "__main__":
    main()
