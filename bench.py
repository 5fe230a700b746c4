"""bench.py -- frames/sec of the DiMP-50 online-tracking hot path on MI355X (BASELINE.json metric).

One "step" = one synthetic tracking frame of BASELINE.json configs[1]:
    classify the 1x512x18x18 test feature with the current 512x4x4 filter -> on-device arg-max -> overwrite one slot
    of the n=50 sample memory (features + box) -> DiMPSteepestDescentGN, 5 iterations, over the 50x512x18x18 memory.
Everything is resident in HBM before the timed region; frames are strictly dependent (frame t+1 classifies with the
filter frame t produced).  One process per GPU, one independent sequence per GPU (weak scaling, no data-path
collective); RCCL is used only for the barrier and the end-of-batch (frames, seconds) gather.

    python bench.py [--gpus N] [--steps K] [--warmup W]          (N > 1: re-executes itself under torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line on rank 0.  Besides the contract fields:
  roofline        dominant feature-pass kernel: algorithmic bytes per launch / its average launch duration.  The
                  duration is rocprofv3's (`--kernel-trace --stats` of this very command in a child process, parsed
                  here -- the same tool that writes profiles/*kernel_stats.csv, so the two agree by construction), taken
                  in the launch mode the headline is timed in: the child REPLAYS a 20-frame hipGraph (`avg_launch_us`,
                  `sum_kernels_us_per_frame.graph` -- comparable with `ms_per_step`); a second child launches the same
                  frames eagerly (`avg_launch_us_eager`, `sum_kernels_us_per_frame.eager`).  Next to them `period_us`:
                  K back-to-back launches of that kernel between ONE HIP event pair on the launch stream (= duration +
                  one dependent-launch boundary).  When rocprofv3 cannot run, `frac` is computed from the event period
                  (pessimistic by the boundary) and `timing` says so.
  repeats         the timed region (exactly --steps frames) run 6 more times AFTER the reported one: us/frame of each,
                  so that the reported value can be placed inside the box's own spread.
  multi_sequence  throughput headroom, NOT the metric: 2 and 4 independent sequences on ONE GPU, each on its own stream
                  with its own state, workspace and graph (what pytracking/evaluation/multi_object_wrapper.py:7-35 would
                  use); the metric's configuration stays one sequence per GPU.
  other_workloads the other BASELINE configs on this GPU (tools/workloads.py): PrDiMP-50 frame, ToMP model prediction, LWL
                  few-shot learner (3 and 4 iterations), ATOM CG update -- ms, algorithmic bytes / flops, roofline fraction.
  end_to_end      the DiMP-50 frame with a stock-PyTorch ResNet-50 (conv1..layer3) in front: backbone ms, frames/s.
  cpu_baseline    the reference's CPU execution path on this host, pinned threads: kind "reference" = the reference's OWN modules
                  (apply_filter + DiMPSteepestDescentGN.forward, imported unmodified from the bundle oracle/_ref that
                  oracle/make_ref_bundle.py writes; oracle/frame_ref.py), with the torch port of the same op sequence
                  (oracle/frame_port.py) beside it under "port" and `one_thread`; kind "port" alone when the bundle is absent.
  gpu_stock_baseline  the same reference modules `.to('cuda')` in stock PyTorch-ROCm (MIOpen grouped convs) on this GPU: what a
                  user of the reference gets on ROCm today without these kernels (SURVEY.md section 8d "Stock-GPU baseline").
"""
import argparse
import contextlib
import csv
import ctypes
import glob
import json
import math
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time

import numpy as np

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from pytracking_amd import _lib, bench_frame, sequences, synth  # noqa: E402
from tools import workloads  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
NUM_ITER = 5
POOL = 50                  # distinct synthetic test features cycled through (one per memory slot)
CPU_THREADS = 16           # pinned at the measured optimum of this path on the 256-core host (profiles/r03a_cpu_thread_scaling.json:
#                            5.4 / 40.8 / 66.2 / 44.9 / 16.4 / 6.2 frames/s at 1 / 8 / 16 / 32 / 64 / 128 threads)


def make_pool(cfg, seed, device):
    rng = np.random.default_rng(seed)
    return torch.from_numpy(synth.clf_features(rng, POOL, cfg["C"], cfg["H"], cfg["W"], cfg["K"])).to(device)


# Frame chains (pt_track_frame_chain_f32, include/pt_hot.h): inside a run of consecutive frames the last filter update of every solve
# is applied in the prologue of the next frame's first correlation instead of by its own dependent launch; the run ends with the
# flush, so every graph / every eager call of run_frames leaves the complete state.  Same bits as the plain frame
# (tests/test_gpu_parity.py::test_frame_chain_deferred_last_update_is_bit_identical).  PT_BENCH_NO_CHAIN=1: the round-5 launch sequence.
CHAIN = os.environ.get("PT_BENCH_NO_CHAIN", "") != "1"
# Closing side of the timed bracket: PT_BENCH_SPIN=1 polls an event recorded behind the K frames before the contract's
# stream.synchronize() + torch.cuda.synchronize() (which then return at once) -- experiment: is the blocking wait's wake-up latency part
# of the 2 ms driver-style region?  (profiles/r06n_*)
SPIN_WAIT = os.environ.get("PT_BENCH_SPIN", "0") == "1"


def run_frames(st, pool, first, count):
    n = st.n
    for f in range(first, first + count):
        st.step(pool[f % POOL], slot=f % n, num_iter=NUM_ITER, defer=CHAIN)
    if CHAIN:
        st.flush()


def reference_available():
    """The reference's own modules importable here?  (/root/reference in the build container; on the GPU box the byte-for-byte
    bundle oracle/_ref/reference that oracle/make_ref_bundle.py writes -- git-ignored, travels with the snapshot.)"""
    try:
        from oracle import frame_ref
        return frame_ref.available()
    except Exception:                                         # noqa: BLE001
        return False


def stock_baseline(cfg, n, device, budget_s, min_frames=10, max_frames=2000, threads=None, impl="port"):
    """The same workload in stock PyTorch on `device`, bounded sample.  impl "reference": the reference's OWN modules
    (`ltr.models.layers.filter.apply_filter` + `DiMPSteepestDescentGN.forward`, optimizer.py:85-170, imported unmodified:
    oracle/frame_ref.ReferenceTracker); impl "port": the torch restatement of that op sequence (oracle/frame_port.TorchCpuTracker).
    Baseline leg only: nothing measured as the product touches oracle/."""
    host = os.cpu_count() or 1
    threads = min(threads or CPU_THREADS, host)
    pool = make_pool(cfg, 99, device)
    if impl == "reference":
        from oracle.frame_ref import ReferenceTracker
        tr = ReferenceTracker(cfg, n, seed=1234, threads=threads, device=device)
        what = ("the reference's own Python (ltr/models/layers/filter.py apply_filter + ltr/models/target_classifier/optimizer.py "
                "DiMPSteepestDescentGN.forward, unmodified bundle oracle/_ref)")
    else:
        from oracle.frame_port import TorchCpuTracker
        tr = TorchCpuTracker(cfg, n, seed=1234, threads=threads, device=device)
        what = ("torch port of the reference's op sequence (grouped F.conv2d apply_filter / feature-as-weights adjoint, 3 passes per "
                "iteration, DistanceMap)")
    sync = torch.cuda.synchronize if str(device).startswith("cuda") else (lambda: None)
    for f in range(3):
        tr.step(pool[f % POOL], f % n, NUM_ITER)
    sync()
    t0 = time.perf_counter()
    frames = 0
    while frames < min_frames or time.perf_counter() - t0 < budget_s:
        tr.step(pool[(frames + 3) % POOL], (frames + 3) % n, NUM_ITER)
        frames += 1
        if frames >= max_frames:
            break
    sync()
    dt = time.perf_counter() - t0
    if str(device).startswith("cuda"):
        return {"value": round(frames / dt, 2), "unit": "frames/s", "kind": impl,
                "sample": f"{frames} frames of the same workload in {dt:.1f}s: {what} in stock PyTorch-ROCm on this GPU, fp32, eager"}
    return {"value": round(frames / dt, 3), "unit": "frames/s", "cores": threads, "host_cores": host, "kind": impl,
            "sample": f"{frames} frames of the same workload in {dt:.1f}s ({what}, fp32, "
                      f"{threads} threads pinned on a {host}-core host)"}


def baselines(cfg, n, dev, want_cpu, want_gpu):
    """cpu_baseline / gpu_stock_baseline objects.  With the reference importable the headline of each is kind "reference" (its own
    modules) and the port sits beside it under "port"; without it kind "port" (and `reference_unavailable` says why)."""
    out = {}
    ref = reference_available()
    if want_cpu:
        if ref:
            cpu = stock_baseline(cfg, n, "cpu", budget_s=12.0, impl="reference")
            port = stock_baseline(cfg, n, "cpu", budget_s=6.0, impl="port")
            cpu["port"] = {k: port[k] for k in ("value", "unit", "cores", "sample")}
            one = stock_baseline(cfg, n, "cpu", budget_s=5.0, min_frames=3, threads=1, impl="reference")
        else:
            cpu = stock_baseline(cfg, n, "cpu", budget_s=12.0, impl="port")
            cpu["reference_unavailable"] = "oracle/_ref/reference not built (python -B oracle/make_ref_bundle.py where /root/reference is mounted)"
            one = stock_baseline(cfg, n, "cpu", budget_s=5.0, min_frames=3, threads=1, impl="port")
        cpu["one_thread"] = {"value": one["value"], "unit": "frames/s", "cores": 1, "kind": one["kind"], "sample": one["sample"]}
        cpu["threads_note"] = ("%d threads = the measured optimum of this path on the 256-core host class "
                               "(profiles/r03a_cpu_thread_scaling.json)" % CPU_THREADS)
        torch.set_num_threads(min(CPU_THREADS, os.cpu_count() or 1))
        out["cpu_baseline"] = cpu
    if want_gpu:
        if ref:
            gpu = stock_baseline(cfg, n, dev, budget_s=4.0, min_frames=50, impl="reference")
            port = stock_baseline(cfg, n, dev, budget_s=3.0, min_frames=50, impl="port")
            gpu["port"] = {k: port[k] for k in ("value", "unit", "sample")}
        else:
            gpu = stock_baseline(cfg, n, dev, budget_s=4.0, min_frames=50, impl="port")
        out["gpu_stock_baseline"] = gpu
    # loud at the TOP level of the line: without the bundle every "reference"-kind figure silently becomes a "port" one
    out["reference_bundle"] = "present" if ref else "ABSENT"
    if not ref:
        out["reference_unavailable"] = ("oracle/_ref/reference not found: cpu_baseline / gpu_stock_baseline are kind \"port\" "
                                        "(python -B oracle/make_ref_bundle.py where /root/reference is mounted)")
    return out


# ---------------------------------------------------------------------------------------------------------------------
# roofline leg
# ---------------------------------------------------------------------------------------------------------------------
def event_period_us(st, stream, which, reps=200):
    """K back-to-back launches of one solver pass between ONE event pair on the launch stream."""
    L = _lib.lib()
    c = st.cfg
    args = (ctypes.byref(st.params), st.filter.data_ptr(), st.mem_feat.data_ptr(), st.mem_bb.data_ptr(),
            st.sample_weight.data_ptr(), st.n, c["C"], c["H"], c["W"], c["K"], NUM_ITER, st.ws.data_ptr(), st.ws.numel(),
            which)
    sp = ctypes.c_void_p(stream.cuda_stream)
    _lib.check(L.pt_track_frame_replay_pass_f32(*args, 20, sp), "pt_track_frame_replay_pass_f32")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    _lib.check(L.pt_track_frame_replay_pass_f32(*args, reps, sp), "pt_track_frame_replay_pass_f32")
    e1.record(stream)
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def stream_floor_us(st, stream, reps=200):
    """Event-pair period of `reps` back-to-back launches of the read-only probe over the sample memory (pt_stream_probe_f32): what the
    memory system delivers for this footprint (the 33 MB stay in the Infinity Cache / L2s between launches, like between passes)."""
    L = _lib.lib()
    scratch = torch.zeros(2048, dtype=torch.float32, device=st.mem_feat.device)
    sp = ctypes.c_void_p(stream.cuda_stream)
    args = (st.mem_feat.data_ptr(), st.mem_feat.numel(), scratch.data_ptr())
    _lib.check(L.pt_stream_probe_f32(*args, 20, sp), "pt_stream_probe_f32")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    _lib.check(L.pt_stream_probe_f32(*args, reps, sp), "pt_stream_probe_f32")
    e1.record(stream)
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


PROF_FRAMES, PROF_WARMUP = 200, 10


PROF_GRAPH = 20            # frames per graph in the graph-replay child (the driver's --steps 20 launch mode)


def rocprof_kernel_stats(workload, frames=PROF_FRAMES, mode="graph"):
    """Run this file's --profile-child leg (mode "graph": replays of one 20-frame hipGraph; "eager": frame by frame) under
    `rocprofv3 --kernel-trace --stats` and return {kernel name: (calls, average ns)} from its kernel-stats CSV, or
    (None, reason)."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    if any(k.startswith(("ROCPROF", "ROCPROFILER")) for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        return None, "this process is itself running under a profiler (its own trace has the kernel durations)"
    tmp = tempfile.mkdtemp(prefix="pt_bench_prof_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [exe, "--kernel-trace", "--stats", "--output-format", "csv", "-d", tmp, "-o", "bench", "--",
           sys.executable, os.path.abspath(__file__), "--profile-child", "--steps", str(frames),
           "--warmup", str(2 if mode == "graph" else PROF_WARMUP),
           "--workload", workload, "--profile-mode", mode]
    try:
        res = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=300)
    except Exception as exc:                                  # noqa: BLE001
        shutil.rmtree(tmp, ignore_errors=True)
        return None, f"rocprofv3 did not run: {exc}"
    files = glob.glob(os.path.join(tmp, "**", "*kernel_stats.csv"), recursive=True)
    dbs = glob.glob(os.path.join(tmp, "**", "*.db"), recursive=True)
    stats = {}
    try:
        if res.returncode == 0 and files:
            with open(files[0], newline="") as fh:
                for row in csv.DictReader(fh):
                    stats[row["Name"]] = (int(row["Calls"]), float(row["AverageNs"]))
        elif res.returncode == 0 and dbs:                     # rocpd (sqlite) output: the same aggregation
            import sqlite3
            con = sqlite3.connect(dbs[0])
            for name, calls, avg in con.execute("select name, count(*), avg(duration) from kernels group by name"):
                stats[name] = (int(calls), float(avg))
            con.close()
    except Exception as exc:                                  # noqa: BLE001
        stats, res = {}, type("R", (), {"returncode": -1, "stderr": str(exc), "stdout": ""})()
    keep = os.environ.get("PT_BENCH_KEEP_STATS")
    if keep and stats:
        with open(keep.replace(".csv", "") + "_" + mode + ".csv", "w") as fh:
            fh.write("Name,Calls,AverageNs\n")
            for nm, (c, a) in sorted(stats.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
                fh.write(f'"{nm}",{c},{a:.1f}\n')
    shutil.rmtree(tmp, ignore_errors=True)
    if not stats:
        return None, f"rocprofv3 exit {res.returncode}: {(res.stderr or res.stdout)[-300:]}"
    return stats, None


def _frames_in_trace(stats):
    """Frames a kernel-stats table covers: the init stage `k_fast_init2` runs exactly once per frame (`k_fast_final` does not since
    round 6: the chained frames fold the final update into the next frame's first correlation and only a flush launches it)."""
    calls = [c for nm, (c, _) in stats.items() if "k_fast_init2" in nm]
    if not calls:
        calls = [c for nm, (c, _) in stats.items() if "k_fast_final" in nm]
    return sum(calls) if calls else None


def roofline(cfg, cfg_name, n, period, stats, why, stats_eager=None, floor_period_us=None):
    """Assemble the roofline object from the kernel-level legs measured in front of the frame timing: `period` (HIP event
    pairs around 200 back-to-back pass launches), `stats` (rocprofv3 kernel averages of the GRAPH-REPLAY child: the mode
    the headline is timed in) and `stats_eager` (the same frames launched one by one)."""
    feat_bytes = 4 * n * cfg["C"] * cfg["H"] * cfg["W"]           # one pass streams the n-sample memory once
    kern = {}
    corr_name = "k_corr2"
    period = {corr_name: period["corr"], "k_adj2": period["adj"]}

    def avg(table, short):
        rows = [(nm, c, a) for nm, (c, a) in (table or {}).items() if short in nm]
        if not rows:
            return None
        calls = sum(c for _, c, _ in rows)
        top = max(rows, key=lambda r: r[1])                                          # the in-iteration instantiation
        return round(sum(c * a for _, c, a in rows) / calls / 1e3, 3), calls, round(top[2] / 1e3, 3), top[0][:60]

    for short in (corr_name, "k_adj2"):
        rec = {"period_us": round(period[short], 3)}
        g, e = avg(stats, short), avg(stats_eager, short)
        if g:
            rec["avg_launch_us"], rec["launches"], rec["avg_launch_us_in_iteration"], rec["instantiation"] = g
        if e:
            rec["avg_launch_us_eager"] = e[0]
            if not g:
                rec["avg_launch_us"], rec["launches"], rec["avg_launch_us_in_iteration"], rec["instantiation"] = e
        dur = rec.get("avg_launch_us", rec["period_us"])
        rec["achieved_GBs"] = round(feat_bytes / dur / 1e3, 1)
        kern[short] = rec
    dom = max(kern, key=lambda k: kern[k].get("avg_launch_us", kern[k]["period_us"]) * kern[k].get("launches", 1))
    traffic, traffic_src = None, None
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")   # committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE pass
    if os.path.exists(pmc):
        rec = json.load(open(pmc)).get(cfg_name, {}).get(dom)
        if rec:
            traffic, traffic_src = rec["hbm_bytes_per_launch"], rec["source"]
    if stats and "avg_launch_us" in kern[dom]:
        timing = ("rocprofv3 --kernel-trace --stats of this command (child process replaying a %d-frame hipGraph, the launch mode "
                  "of the headline), parsed in-process" % PROF_GRAPH)
    elif stats_eager and "avg_launch_us" in kern[dom]:
        timing = "rocprofv3 --kernel-trace --stats of this command (child process, EAGER launches: the graph child failed: %s)" % why
    else:
        timing = f"HIP event pair around 200 back-to-back launches (includes one launch boundary each); {why}"

    def per_frame(table):
        fr = _frames_in_trace(table) if table else None
        return None if not fr else round(sum(c * a for nm, (c, a) in table.items() if "k_stream_probe" not in nm) / 1e3 / fr, 2)
    # the streaming floor of a pass: the read-only probe over the same memory, by the same clock as `avg_launch_us` when the trace has
    # it (rocprofv3 duration of k_stream_probe in the graph child), else by the event-pair period
    probe = avg(stats, "k_stream_probe") or avg(stats_eager, "k_stream_probe")
    floor_us = probe[0] if probe else (round(floor_period_us, 3) if floor_period_us else None)
    dom_us = kern[dom].get("avg_launch_us", kern[dom]["period_us"])
    floor = None
    if floor_us:
        floor = {"stream_floor_us": floor_us, "stream_floor_GBs": round(feat_bytes / floor_us / 1e3, 1),
                 "stream_floor_period_us": None if floor_period_us is None else round(floor_period_us, 3),
                 "frac_of_floor": round(floor_us / dom_us, 4),
                 "what": "a kernel that only reads the same sample memory (2048 workgroups, 16-byte loads, 8 in flight per lane; "
                         "pt_stream_probe_f32): the footprint lives in the Infinity Cache / L2s between launches, so this -- not 8 TB/s -- "
                         "is what the memory system can deliver to a pass; frac_of_floor = stream_floor_us / the dominant pass's duration"}
    return {"bound": "hbm", "kernel": dom, "achieved": kern[dom]["achieved_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(kern[dom]["achieved_GBs"] / HBM_PEAK_GBS, 4), "stream_floor_us": floor_us if floor_us else None,
            "frac_of_floor": floor["frac_of_floor"] if floor else None, "stream_floor": floor,
            "traffic": traffic, "traffic_source": traffic_src,
            "avg_launch_us": kern[dom].get("avg_launch_us", kern[dom]["period_us"]), "timing": timing,
            "algorithmic_bytes_per_launch": feat_bytes, "kernels": kern,
            "sum_kernels_us_per_frame": {"graph": per_frame(stats), "eager": per_frame(stats_eager),
                                         "note": "sum over ALL kernels of the trace / frames in it; each rocprofv3 duration carries the "
                                                 "kernel's own start-up and drain, the frame time (ms_per_step) additionally the "
                                                 "dependent-launch boundaries between its 17 kernels (18 without the chained final update)"}}


def multi_sequence(cfg, cfg_name, n, dev, counts=(2, 4), G=20, replays=10):
    """Throughput headroom (extra key, not the metric): S independent sequences on one GPU, each with its own TrackState
    (filter, memory, workspace), stream and 20-frame hipGraph; all S graphs are replayed `replays` times, interleaved."""
    out = {}
    for S in counts:
        streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
        states, graphs = [], []
        for k, sm in enumerate(streams):
            with torch.cuda.stream(sm):
                st = bench_frame.TrackState(cfg, n, seed=4000 + k, device=dev, kind="dimp" if cfg_name == "dimp50" else "prdimp")
                pool = make_pool(cfg, 5000 + k, dev)
                run_frames(st, pool, 0, 2)
                sm.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=sm):
                    run_frames(st, pool, 2, G)
                g.replay()
                sm.synchronize()
                states.append((st, pool)); graphs.append(g)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(replays):
            for sm, g in zip(streams, graphs):
                with torch.cuda.stream(sm):
                    g.replay()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out[str(S)] = {"frames_per_s": round(S * G * replays / dt, 1), "us_per_frame_per_sequence": round(1e6 * dt / (G * replays), 2)}
        del states, graphs, streams
    out["what"] = ("S independent sequences on ONE GPU, one stream + state + workspace + %d-frame hipGraph each, %d interleaved "
                   "replays; aggregate frames/s over the S sequences.  Headroom only: the metric is one sequence per GPU" % (G, replays))
    return out


def head_inclusive(cfg, n, dev, stream, frames=200):
    """Second number: the same frame with the classification-feature head (conv 1024 -> C, 3x3,
    InstanceL2Norm) in front, its output written straight into the memory slot (pt_track_frame_head_f32); hipGraph of 25
    frames replayed.  The ResNet-50 backbone is stock PyTorch and not part of either number."""
    st = bench_frame.TrackState(cfg, n, seed=777, device=dev)
    rng = np.random.default_rng(778)
    w = torch.from_numpy(rng.standard_normal((cfg["C"], 1024, 3, 3), dtype=np.float32) * np.float32(0.02)).to(dev)
    st.attach_head(w, math.sqrt(1.0 / (cfg["C"] * cfg["K"] ** 2)))
    xb = torch.from_numpy(rng.standard_normal((8, 1024, cfg["H"], cfg["W"]), dtype=np.float32)).to(dev)
    G = 25
    with torch.cuda.stream(stream):
        for f in range(2):
            st.step_from_backbone(xb[f % 8], f % n, NUM_ITER)
        stream.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            for f in range(G):
                st.step_from_backbone(xb[f % 8], f % n, NUM_ITER)
        g.replay()
        stream.synchronize()
        t0 = time.perf_counter()
        for _ in range(frames // G):
            g.replay()
        stream.synchronize()
        dt = time.perf_counter() - t0
    fr = (frames // G) * G
    return {"value": round(fr / dt, 2), "unit": "frames/s", "us_per_frame": round(1e6 * dt / fr, 2),
            "what": "classification-feature head (1024->%d, 3x3, InstanceL2Norm) writing the memory slot + the frame above; "
                    "hipGraph replay, 25 frames per graph" % cfg["C"]}


class _DryStream:
    """Control-flow dry run (PT_BENCH_DRYRUN=1): stands in for a HIP stream on a box without a GPU."""
    cuda_stream = 0

    def synchronize(self):
        pass


class _DryState:
    """Control-flow dry run: stands in for bench_frame.TrackState.  `step` does a token amount of host work so that the timed
    region has a duration; nothing about it is a measurement."""

    def __init__(self, cfg, n, seed):
        self.cfg, self.n = dict(cfg), n
        self.acc = torch.zeros(64, dtype=torch.float64) + seed

    def step(self, feat, slot, num_iter, defer=False):
        self.acc = (self.acc * 1.0000001 + slot + num_iter).sin_()

    def flush(self):
        pass

    def bytes_per_solve(self, num_iter):
        c = self.cfg
        return 2 * num_iter * 4 * self.n * c["C"] * c["H"] * c["W"]


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--no-graph", action="store_true", help="launch every frame eagerly instead of replaying hipGraphs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-other", action="store_true", help="skip the other BASELINE workloads and the end-to-end leg")
    ap.add_argument("--no-clock-warmup", action="store_true",
                    help="skip the ~40 ms of idempotent pass replays in front of the warm-up frames (rounds 1-3 protocol)")
    ap.add_argument("--profile-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--profile-mode", default="graph", choices=("graph", "eager"), help=argparse.SUPPRESS)
    ap.add_argument("--workload", default="dimp50", choices=("dimp50", "prdimp50"),
                    help="dimp50 = BASELINE configs[1] (the metric's configuration); prdimp50 = configs[2]'s per-GPU workload")
    args = ap.parse_args()

    # PT_BENCH_DRYRUN=1: walk this file's control flow (launcher re-exec, rank wiring, build-on-rank-0 + barrier, warm-up, barrier-
    # bracketed timed region, closing barrier, gather, ONE JSON line on rank 0) on a box WITHOUT a GPU: gloo instead of RCCL, a stub
    # sequence state, no graphs, no kernel-level legs.  tests/test_bench_multirank_cpu.py runs it with 2 and 8 ranks; the line it
    # prints says "data": "dry-run" and is not a measurement.
    dry = os.environ.get("PT_BENCH_DRYRUN") == "1"
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        if "WORLD_SIZE" in os.environ:
            raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} ranks")
        # started as plain `python bench.py --gpus N`: become the N-rank job (one process per GPU over RCCL)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        os.execv(sys.executable, cmd)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # stdout carries ONE JSON line and nothing else: everything any library writes to descriptor 1 from here on (RCCL prints a
    # version banner there from its C runtime, flushed at exit -- i.e. BEHIND the line) goes to stderr; the line itself is written
    # to the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if dry:
        dev = torch.device("cpu")
        args.no_roofline = args.no_other = args.no_cpu_baseline = args.no_gpu_baseline = args.no_graph = True
    else:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    sync_all = (lambda: None) if dry else torch.cuda.synchronize
    dist = None
    if world > 1 or "RANK" in os.environ:                      # under a launcher even one rank takes the RCCL path
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if dry:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)

    if _lib.needs_build():
        if rank == 0:
            _lib.build_library()
        if dist is not None:
            dist.barrier()
    cfg_name = args.workload
    cfg = synth.DIMP50 if cfg_name == "dimp50" else synth.PRDIMP50
    n = cfg["memory"]
    if dry:
        st, pool = _DryState(cfg, n, seed=1234 + rank), torch.zeros(POOL, 1)
        stream, on_stream = _DryStream(), (lambda s_: contextlib.nullcontext())
    else:
        st = bench_frame.TrackState(cfg, n, seed=1234 + rank, device=dev,      # one independent sequence per GPU
                                    kind="dimp" if cfg_name == "dimp50" else "prdimp")
        pool = make_pool(cfg, 4321 + rank, dev)
        stream, on_stream = torch.cuda.Stream(device=dev), torch.cuda.stream
    K, Wm = args.steps, args.warmup

    if args.profile_child:                                     # the legs rocprofv3 traces, nothing else in the process
        with torch.cuda.stream(stream):
            if args.profile_mode == "eager":
                run_frames(st, pool, 0, Wm + K)
            else:                                              # the headline's launch mode: replays of one 20-frame graph
                run_frames(st, pool, 0, Wm)
                stream.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=stream):
                    run_frames(st, pool, Wm, PROF_GRAPH)
                for _ in range(max(1, K // PROF_GRAPH)):
                    g.replay()
            stream_floor_us(st, stream, reps=40)               # the read-only probe, so that the same trace times it (k_stream_probe)
            stream.synchronize()
        return

    # ---- kernel-level legs first (rocprofv3 child process, then the event-pair periods of the two passes in this process):
    #      the frame timing that follows starts on a device that has just been busy, whatever --warmup says
    want_roof = not args.no_roofline and rank == 0
    # the profiling children only in a single-rank run (the other ranks of a multi-GPU job would idle at the barrier for ~25 s);
    # a multi-rank line carries the roofline of rank 0 from the event-pair periods
    prof = want_roof and world == 1
    stats, why = rocprof_kernel_stats(cfg_name, mode="graph") if prof else (None, "multi-rank run: no profiling child" if want_roof else "not requested")
    stats_eager = rocprof_kernel_stats(cfg_name, frames=60, mode="eager")[0] if prof else None

    # ---- frame launcher: hipGraphs of G consecutive frames starting at frame --warmup, G = the timed steps themselves when
    #      they fit one memory cycle, else their common divisor with the memory size (whole graph replays; a graph is valid
    #      for one start slot, so one is captured per distinct start slot; every memory slot keeps being overwritten in
    #      turn); the warm-up frames as one more graph.  Eager launches when G < 5 / --no-graph.
    G = K if K <= n else math.gcd(K, n)
    use_graph = (not args.no_graph) and G >= 5
    graphs = {}
    with on_stream(stream):
        run_frames(st, pool, 0, 2)                             # first-touch / code-object load outside everything
        stream.synchronize()

        def capture(first, count):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=stream):
                run_frames(st, pool, first, count)
            return g

        if use_graph:
            for j in range(K // G):                            # graph keyed by its start slot
                s0 = (Wm + j * G) % n
                if s0 not in graphs:
                    graphs[s0] = capture(s0, G)
            warm_graph = capture(0, Wm) if Wm >= 5 else None

        def advance(first, count):
            f, end = first, first + count
            while f < end:
                if use_graph and (f - Wm) % G == 0 and f >= Wm and end - f >= G:
                    graphs[f % n].replay()
                    f += G
                else:
                    run_frames(st, pool, f, 1)
                    f += 1

        # Clock warm-up, every rank, in FRONT of the W warm-up frames: the two feature passes of the last solve re-issued back to back
        # for ~40 ms (pt_track_frame_replay_pass_f32: idempotent, the sequence state does not move).  After the idle seconds of the
        # profiling children the first 2 ms burst ran 2-3 % slower than the following ones and settled over ~3 replays of a
        # 20-frame graph (profiles/r04c_short_region_warmup.txt: 112.1 -> 110.5 -> 109.1 us/frame; hipGraphUpload changed
        # nothing, an untimed first launch of the executable recovered a third).  The same launches are the roofline leg's
        # event-pair periods (last round trip kept).  The W warm-up frames then run directly in front of the timed region.
        period = None
        try:
            for _ in range(0 if (dry or args.no_clock_warmup) else 12):                                # ~40 ms: the ramp was still visible after 10 (repeats 109.4 -> 108.4 us)
                period = {"corr": event_period_us(st, stream, 0), "adj": event_period_us(st, stream, 1)}
        except RuntimeError:
            period = None                                      # configuration outside the fast path: no replay helper
        if use_graph and warm_graph is not None:
            warm_graph.replay()
        else:
            advance(0, Wm)
        stream.synchronize()
        if dist is not None:
            dist.barrier()
        sync_all()
        t0 = time.perf_counter()
        advance(Wm, K)
        if SPIN_WAIT and not dry:                              # poll for completion, THEN the contract's synchronize (returns at once)
            done = torch.cuda.Event()
            done.record(stream)
            while not done.query():
                pass
        stream.synchronize()
        sync_all()
        elapsed = time.perf_counter() - t0                     # this rank's K frames; the job's time is the MAX over ranks (gather below)
        if dist is not None:
            dist.barrier()                                     # closing barrier of the bracket: not part of anybody's K frames
        elapsed_closed = time.perf_counter() - t0              # rounds 1-3 bracket: this rank's clock incl. the closing barrier

        # the same K-step region again (state keeps advancing; same graphs): where the reported value sits in this box's spread
        repeats = []
        if rank == 0 and world == 1 and not args.no_roofline:
            for _ in range(6):
                if use_graph and warm_graph is not None:
                    warm_graph.replay()
                else:
                    advance(0, Wm)
                stream.synchronize()
                t1 = time.perf_counter()
                advance(Wm, K)
                if SPIN_WAIT and not dry:
                    done = torch.cuda.Event()
                    done.record(stream)
                    while not done.query():
                        pass
                stream.synchronize()
                repeats.append(round(1e6 * (time.perf_counter() - t1) / K, 2))

        floor_period = None
        if want_roof and period:
            try:
                floor_period = stream_floor_us(st, stream)
            except RuntimeError:
                floor_period = None
        roof = roofline(cfg, cfg_name, n, period, stats, why, stats_eager, floor_period) if want_roof and period else None

    # the only collective: the end-of-batch (frames, seconds) gather; whole-job rate = all frames / slowest rank
    total_frames, tmax, per_rank = sequences.gather_throughput(K, elapsed, device=dev)
    value = total_frames / tmax

    if rank == 0:
        solve_bytes = st.bytes_per_solve(NUM_ITER)
        if roof is not None:
            gbs = solve_bytes * (K / tmax) / 1e9
            roof["solve_level"] = {"algorithmic_bytes_per_frame": solve_bytes, "achieved_GBs": round(gbs, 1),
                                   "frac": round(gbs / HBM_PEAK_GBS, 4),
                                   "note": "2 feature reads per iteration x 5 iterations (SURVEY 8d) / measured frame time"}
        launch = (f"hipGraph replay, {G} frames per graph ({K // G} replay(s) in the timed region, {len(graphs)} start slot(s)"
                  + ("" if args.no_clock_warmup else "; clock warm-up: ~40 ms of idempotent pass replays in front of the warm-up frames")
                  + ")" if use_graph
                  else "eager (18 launches per frame)")
        if CHAIN:
            launch += ("; frame chain: the last filter update of each solve rides on the next frame's first correlation "
                       "(17 launches per frame + one flush per run of frames)")
        out = {
            "metric": "frames/sec DiMP-50 online track (288x288, 5 SD iters)" if cfg_name == "dimp50" else "frames/sec PrDiMP-50 online track (352x352, 5 SD iters)", "value": round(value, 2),
            "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": Wm,
            "ms_per_step": round(1e3 * tmax / K, 5), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "dry-run (no GPU: control flow only, NOT a measurement)" if dry else "synthetic",
            "config": {"workload": ("BASELINE configs[1]: DiMP-50 single sequence per GPU; per frame classify(1x512x18x18) "
                                    "+ arg-max + memory insert + DiMPSteepestDescentGN(5 it) over n=50x512x18x18, K=4")
                       if cfg_name == "dimp50" else
                       ("BASELINE configs[2] per-GPU workload: PrDiMP-50 single sequence per GPU; per frame "
                        "classify(1x512x22x22) + arg-max + memory insert + PrDiMPSteepestDescentNewton(5 it) over "
                        "n=50x512x22x22, K=4"),
                       "sequences_per_gpu": 1, "launch": launch, "parallelism": f"{world} independent sequences"},
            "roofline": roof,
            # self-verifying multi-GPU record: what every rank timed, and how many ranks the collective really had
            "per_rank": [{"rank": r, "frames": f, "seconds": round(sec, 6), "frames_per_s": round(f / sec, 1)}
                         for r, (f, sec) in enumerate(per_rank)],
            # the two brackets side by side: `value` uses the MAX over ranks of each rank's own K frames (closing barrier
            # outside); rounds 1-3 timed rank 0 up to and including the closing barrier
            "bracket": {"max_rank_seconds": round(tmax, 6), "rank0_seconds_incl_closing_barrier": round(elapsed_closed, 6),
                        "value_incl_closing_barrier": round(total_frames / elapsed_closed, 2),
                        "clock_warmup": not (dry or args.no_clock_warmup)},
            "collective": {"backend": (dist.get_backend() if dist is not None else None),
                           "ranks": (dist.get_world_size() if dist is not None else 1),
                           "what": "barrier + one 16-byte all_gather of (frames, seconds); RCCL when backend == nccl"},
        }
        if repeats:
            out["repeats"] = {"us_per_frame": repeats, "reported_us_per_frame": round(1e6 * tmax / K, 2),
                              "what": "the timed region (exactly --steps frames) run 6 more times after the reported one"}
        if world == 1 and not args.no_roofline:
            try:
                out["multi_sequence"] = multi_sequence(cfg, cfg_name, n, dev)
            except Exception as exc:                             # noqa: BLE001
                out["multi_sequence"] = {"error": f"{type(exc).__name__}: {exc}"[:300]}
        if world == 1 and cfg_name == "dimp50" and not args.no_roofline:
            out["head_inclusive"] = head_inclusive(cfg, n, dev, stream)
        if world == 1 and cfg_name == "dimp50":
            if not args.no_other:
                out["other_workloads"] = workloads.all_other(dev)
            out.update(baselines(cfg, n, dev, not args.no_cpu_baseline, not args.no_gpu_baseline))
            if not args.no_other:                                # last: its MIOpen find mode must not touch the baseline above
                try:
                    out["end_to_end"] = workloads.end_to_end(dev)
                except Exception as exc:                         # noqa: BLE001
                    out["end_to_end"] = {"error": f"{type(exc).__name__}: {exc}"[:300]}
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())      # the one line on the real stdout
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
