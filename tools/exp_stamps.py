"""Experiment: phase time stamps of the two solver passes (library built with -DPT_STAMPS, selected via PT_HOT_LIB).
   PT_HOT_LIB=.../libpt_hot_stamps.so python tools/exp_stamps.py [n ...]"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from pytracking_amd import _lib, bench_frame, synth  # noqa: E402

if os.environ.get("PT_STAMP_SET") == "2":
    ADJ = ["entry", "small loads requested", "feature loads requested", "maps zeroed", "table written", "barrier 1", "update stage done", "end"]
else:
    ADJ = None
NAMES = {0: ["entry", "loads issued", "filter reduced+staged", "barrier 1", "MFMA loop done", "T written", "barrier 2", "end"],
         1: ["entry", "tables + update inputs requested", "barrier 1", "update stage done", "barrier 2", "MFMA loop done",
             "red written + barrier 3", "end"]}


def main():
    ns = [int(a) for a in sys.argv[1:]] or [32, 50]
    cfg = synth.DIMP50
    dev = torch.device("cuda:0")
    L = _lib.lib()
    L.pt_debug_set_stamps.argtypes = [ctypes.c_void_p]
    stream = torch.cuda.Stream()
    pool = bench.make_pool(cfg, 99, dev)
    with torch.cuda.stream(stream):
        for n in ns:
            st = bench_frame.TrackState(cfg, n, seed=1234, device=dev)
            bench.run_frames(st, pool, 0, 10)
            stream.synchronize()
            c = st.cfg
            for which in (0, 1):
                args = (ctypes.byref(st.params), st.filter.data_ptr(), st.mem_feat.data_ptr(), st.mem_bb.data_ptr(),
                        st.sample_weight.data_ptr(), st.n, c["C"], c["H"], c["W"], c["K"], bench.NUM_ITER, st.ws.data_ptr(),
                        st.ws.numel(), which)
                sp = ctypes.c_void_p(stream.cuda_stream)
                nwg = 8 * n if which == 0 else 256 * 2
                buf = torch.zeros(nwg * 16 * 8, dtype=torch.int64, device=dev)
                L.pt_track_frame_replay_pass_f32(*args, 20, sp)                # warm
                acc = []
                for rep in range(5):
                    buf.zero_()
                    stream.synchronize()
                    L.pt_debug_set_stamps(ctypes.c_void_p(buf.data_ptr()))
                    L.pt_track_frame_replay_pass_f32(*args, 1, sp)
                    stream.synchronize()
                    L.pt_debug_set_stamps(None)
                    t = buf.cpu().numpy().reshape(nwg, 16, 8).astype(np.float64)
                    t[t == 0] = np.nan
                    t = (t - np.nanmin(t[:, :, 0])) * 0.01                      # 100 MHz -> us
                    acc.append(t)
                t = np.nanmean(np.stack(acc), axis=0)
                print(f"n={n} {'k_corr2 (fused)' if which == 0 else 'k_adj2 (sd)'}: per-stamp over all waves [us since first wave entry]: min / mean / max"
                      f" | per-WORKGROUP last wave: mean")
                for k in range(8):
                    col = t[:, :, k]
                    if np.all(np.isnan(col)):
                        continue
                    wgmax = np.nanmax(col, axis=1)
                    print(f"   {k} {NAMES[which][k]:34s} {np.nanmin(col):6.2f} {np.nanmean(col):6.2f} {np.nanmax(col):6.2f}   | {np.nanmean(wgmax):6.2f}")
                dur = np.nanmax(t[:, :, 7], axis=1) - np.nanmin(t[:, :, 0], axis=1)
                print(f"   workgroup duration (entry of first wave -> end of last): min {np.nanmin(dur):.2f} mean {np.nanmean(dur):.2f} max {np.nanmax(dur):.2f}")


if ADJ:
    NAMES[1] = ADJ

if __name__ == "__main__":
    main()
