"""bench.py's JSON assembly, without a GPU: the roofline object is computed from kernel-stat tables and event periods by plain
Python (bench.roofline); the committed counter summary must be this round's."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _stats(corr_ns, adj_ns, frames):
    return {"void k_corr2<16, true, 8, true, 1>(float const*, long)": (5 * frames, corr_ns),
            "void k_corr2<16, true, 0, true, 1>(float const*, long)": (frames, corr_ns - 500.0),
            "void k_adj2<1, 6, 16, false>(float const*, long)": (5 * frames, adj_ns),
            "k_fast_sgq2(float const*)": (5 * frames, 5000.0), "k_fast_init2(float const*)": (frames, 5000.0),
            "k_fast_final(float const*)": (frames, 5000.0)}


def test_roofline_object_from_graph_and_eager_tables():
    import bench
    from pytracking_amd import synth
    cfg, n = synth.DIMP50, 50
    r = bench.roofline(cfg, "dimp50", n, {"corr": 8.1, "adj": 7.6}, _stats(8200.0, 7600.0, 200), "", _stats(8700.0, 7900.0, 70))
    feat_bytes = 4 * n * cfg["C"] * cfg["H"] * cfg["W"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert r["kernel"] == "k_corr2" and r["algorithmic_bytes_per_launch"] == feat_bytes == 33177600
    k = r["kernels"]["k_corr2"]
    assert k["launches"] == 1200 and k["avg_launch_us_in_iteration"] == 8.2 and k["avg_launch_us_eager"] > k["avg_launch_us"]
    assert abs(r["avg_launch_us"] - (5 * 8.2 + 7.7) / 6) < 2e-3                     # all instantiations, graph-replay table
    assert abs(r["achieved"] - feat_bytes / r["avg_launch_us"] / 1e3) < 0.1 and abs(r["frac"] - r["achieved"] / 8000.0) < 1e-4
    assert "hipGraph" in r["timing"]
    per = r["sum_kernels_us_per_frame"]
    want = (5 * 8.2 + 7.7 + 5 * 7.6 + 7 * 5.0)
    assert abs(per["graph"] - want) < 0.02 and per["eager"] > per["graph"]
    # the streaming floor: the probe's duration from the same trace when it is there, the event period otherwise
    st = _stats(8200.0, 7600.0, 200)
    st["k_stream_probe(float const*, long, float*)"] = (40, 3700.0)
    r3 = bench.roofline(cfg, "dimp50", n, {"corr": 8.1, "adj": 7.6}, st, "", None, floor_period_us=4.4)
    assert r3["stream_floor_us"] == 3.7 and abs(r3["frac_of_floor"] - 3.7 / r3["avg_launch_us"]) < 1e-3
    assert r3["stream_floor"]["stream_floor_period_us"] == 4.4 and r3["stream_floor"]["stream_floor_GBs"] > 8000
    assert abs(r3["sum_kernels_us_per_frame"]["graph"] - want) < 0.02          # the probe launches are not part of a frame
    # chained frames (round 6): k_fast_final only in the flushes -- the frame count comes from the init stage
    ch = _stats(8200.0, 7600.0, 200)
    ch["k_fast_final(float const*)"] = (10, 5000.0)
    r5 = bench.roofline(cfg, "dimp50", n, {"corr": 8.1, "adj": 7.6}, ch, "", None)
    assert abs(r5["sum_kernels_us_per_frame"]["graph"] - (want - 5.0 + 10 * 5.0 / 200)) < 0.02
    r4 = bench.roofline(cfg, "dimp50", n, {"corr": 8.1, "adj": 7.6}, None, "x", None, floor_period_us=4.4)
    assert r4["stream_floor_us"] == 4.4 and r["stream_floor_us"] is None
    # no trace at all: the event period (pessimistic by one launch boundary) carries the fraction and `timing` says so
    r2 = bench.roofline(cfg, "dimp50", n, {"corr": 8.1, "adj": 7.6}, None, "multi-rank run: no profiling child", None)
    assert r2["avg_launch_us"] == 8.1 and "event pair" in r2["timing"] and r2["sum_kernels_us_per_frame"]["graph"] is None


def test_committed_counter_summary_is_this_rounds():
    """`roofline.traffic` comes from profiles/pmc_traffic.json, which tools/pmc_traffic.py writes from the counter file of the
    CURRENT kernels: the source must name a round-6 file and the instantiations the passes run now (sample pairs: NK = 16, one
    wave per tile; PrDiMP's 24-group adjoint)."""
    rec = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    for wl, kern, inst in (("dimp50", "k_corr2", "k_corr2<16, true, 8, true, 1>"), ("dimp50", "k_adj2", "k_adj2<1, 6, 16, false>"),
                           ("prdimp50", "k_corr2", "k_corr2<16, false, 8, true, 1>"), ("prdimp50", "k_adj2", "k_adj2<4, 9, 24, false>")):
        r = rec[wl][kern]
        assert "profiles/r06" in r["source"] and inst in r["source"], r["source"]
        assert os.path.exists(os.path.join(ROOT, r["source"].split(":")[0]))
    alg = 4 * 50 * 512 * 18 * 18
    assert 1.0 <= rec["dimp50"]["k_corr2"]["hbm_bytes_per_launch"] / alg < 1.1       # traffic = 1.03 x the algorithmic bytes
