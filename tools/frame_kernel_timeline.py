"""Average GPU timeline of the frames in a `rocprofv3 --kernel-trace --output-format csv` run of a frame loop: for the k-th kernel of a
frame (frames end with k_iou_final) the mean start offset, duration and gap to its predecessor over all complete frames.
    python tools/frame_kernel_timeline.py DIR [skip_first_frames]"""
import csv
import glob
import os
import sys


def main():
    f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)
    skip = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    rows = sorted(csv.DictReader(open(f[0], newline="")), key=lambda r: int(r["Start_Timestamp"]))
    frames, cur = [], []
    for r in rows:
        name = r["Kernel_Name"]
        short = name.split("<")[0].split("(anonymous namespace)::")[-1].split("(")[0].replace("void ", "")
        cur.append((short, int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
        if "k_iou_final" in name:
            frames.append(cur)
            cur = []
    frames = frames[skip:]
    if not frames:
        print("no frames")
        return
    from collections import Counter
    L = Counter(len(fr) for fr in frames).most_common(1)[0][0]
    frames = [fr for fr in frames if len(fr) == L]
    n = len(frames)
    print(f"{n} frames of {L} kernels; columns: kernel, mean start offset us, mean duration us, mean gap to predecessor us")
    tot_d = tot_g = 0.0
    for k in range(L):
        st = sum(fr[k][1] - fr[0][1] for fr in frames) / n / 1e3
        du = sum(fr[k][2] - fr[k][1] for fr in frames) / n / 1e3
        gp = 0.0 if k == 0 else sum(fr[k][1] - fr[k - 1][2] for fr in frames) / n / 1e3
        tot_d += du
        tot_g += gp
        print(f"{frames[0][k][0]:28s} {st:8.1f} {du:7.2f} {gp:7.2f}")
    span = sum(fr[-1][2] - fr[0][1] for fr in frames) / n / 1e3
    period = sum(frames[i + 1][0][1] - frames[i][0][1] for i in range(n - 1)) / max(n - 1, 1) / 1e3
    print(f"sum durations {tot_d:.1f} us, sum gaps {tot_g:.1f} us, first start -> last end {span:.1f} us, frame period {period:.1f} us")


if __name__ == "__main__":
    main()
