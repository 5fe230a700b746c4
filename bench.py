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
                  here -- the same tool that writes profiles/*kernel_stats.csv, so the two agree by construction); next
                  to it `period_us`: K back-to-back launches of that kernel between ONE HIP event pair on the launch
                  stream (= duration + one dependent-launch boundary).  When rocprofv3 cannot run, `frac` is computed
                  from the event period (pessimistic by the boundary) and `timing` says so.
  other_workloads the other BASELINE configs on this GPU (tools/workloads.py): PrDiMP-50 frame, ToMP model prediction, LWL
                  few-shot learner (3 and 4 iterations), ATOM CG update -- ms, algorithmic bytes / flops, roofline fraction.
  end_to_end      the DiMP-50 frame with a stock-PyTorch ResNet-50 (conv1..layer3) in front: backbone ms, frames/s.
  cpu_baseline    the reference's CPU execution path (torch-CPU port, oracle/frame_port.py) on this host, pinned threads
                  (+ `one_thread`: the same on a single thread).
  gpu_stock_baseline  the same stock-PyTorch op sequence (MIOpen grouped convs) on this GPU: what a user of the reference
                  gets on ROCm today without these kernels (SURVEY.md section 8d "Stock-GPU baseline").
"""
import argparse
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


def run_frames(st, pool, first, count):
    n = st.n
    for f in range(first, first + count):
        st.step(pool[f % POOL], slot=f % n, num_iter=NUM_ITER)


def stock_baseline(cfg, n, device, budget_s, min_frames=10, max_frames=2000, threads=None):
    """The reference's op sequence in stock PyTorch (oracle/frame_port.TorchCpuTracker) on `device`, same workload,
    bounded sample.  Baseline leg only: nothing measured as the product touches oracle/."""
    from oracle.frame_port import TorchCpuTracker
    host = os.cpu_count() or 1
    threads = min(threads or CPU_THREADS, host)
    pool = make_pool(cfg, 99, device)
    tr = TorchCpuTracker(cfg, n, seed=1234, threads=threads, device=device)
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
        return {"value": round(frames / dt, 2), "unit": "frames/s", "kind": "port",
                "sample": f"{frames} frames of the same workload in {dt:.1f}s: the reference's op sequence (grouped "
                          f"F.conv2d apply_filter / feature-as-weights adjoint, 3 passes per iteration, DistanceMap) in "
                          f"stock PyTorch-ROCm on this GPU, fp32, eager"}
    return {"value": round(frames / dt, 3), "unit": "frames/s", "cores": threads, "host_cores": host, "kind": "port",
            "sample": f"{frames} frames of the same workload in {dt:.1f}s (torch-CPU port of the reference path, fp32, "
                      f"{threads} threads pinned on a {host}-core host)"}


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


PROF_FRAMES, PROF_WARMUP = 120, 10


def rocprof_kernel_stats(workload, frames=PROF_FRAMES):
    """Run this file's --profile-child leg under `rocprofv3 --kernel-trace --stats` and return {kernel name: (calls,
    average ns)} from its kernel-stats CSV, or (None, reason)."""
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
           sys.executable, os.path.abspath(__file__), "--profile-child", "--steps", str(frames), "--warmup", str(PROF_WARMUP),
           "--workload", workload]
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
        with open(keep, "w") as fh:
            fh.write("Name,Calls,AverageNs\n")
            for nm, (c, a) in sorted(stats.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
                fh.write(f'"{nm}",{c},{a:.1f}\n')
    shutil.rmtree(tmp, ignore_errors=True)
    if not stats:
        return None, f"rocprofv3 exit {res.returncode}: {(res.stderr or res.stdout)[-300:]}"
    return stats, None


def roofline(cfg, cfg_name, n, period, stats, why):
    """Assemble the roofline object from the two kernel-level legs measured in front of the frame timing: `period` (HIP
    event pairs around 200 back-to-back pass launches) and `stats` (rocprofv3 kernel averages of the child process)."""
    feat_bytes = 4 * n * cfg["C"] * cfg["H"] * cfg["W"]           # one pass streams the n-sample memory once
    kern = {}
    corr_name = "k_corr2"
    period = {corr_name: period["corr"], "k_adj2": period["adj"]}
    for short in (corr_name, "k_adj2"):
        rec = {"period_us": round(period[short], 3)}
        if stats:
            rows = [(nm, c, a) for nm, (c, a) in stats.items() if short in nm]
            if rows:
                calls = sum(c for _, c, _ in rows)
                rec["avg_launch_us"] = round(sum(c * a for _, c, a in rows) / calls / 1e3, 3)    # all instantiations
                rec["launches"] = calls
                fused = [(c, a) for nm, c, a in rows if c == max(r[1] for r in rows)]             # the in-iteration one
                rec["avg_launch_us_in_iteration"] = round(fused[0][1] / 1e3, 3)
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
    timing = ("rocprofv3 --kernel-trace --stats of this command (child process), parsed in-process" if stats and
              "avg_launch_us" in kern[dom] else f"HIP event pair around 200 back-to-back launches (includes one launch boundary each); {why}")
    return {"bound": "hbm", "kernel": dom, "achieved": kern[dom]["achieved_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(kern[dom]["achieved_GBs"] / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
            "avg_launch_us": kern[dom].get("avg_launch_us", kern[dom]["period_us"]), "timing": timing,
            "algorithmic_bytes_per_launch": feat_bytes, "kernels": kern,
            "all_kernels_us_per_frame": None if not stats else round(sum(c * a for c, a in stats.values()) / 1e3 / (PROF_FRAMES + PROF_WARMUP), 2)}


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
    ap.add_argument("--profile-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--workload", default="dimp50", choices=("dimp50", "prdimp50"),
                    help="dimp50 = BASELINE configs[1] (the metric's configuration); prdimp50 = configs[2]'s per-GPU workload")
    args = ap.parse_args()

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
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or "RANK" in os.environ:                      # under a launcher even one rank takes the RCCL path
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend="nccl", device_id=dev)

    if _lib.needs_build():
        if rank == 0:
            _lib.build_library()
        if dist is not None:
            dist.barrier()
    cfg_name = args.workload
    cfg = synth.DIMP50 if cfg_name == "dimp50" else synth.PRDIMP50
    n = cfg["memory"]
    st = bench_frame.TrackState(cfg, n, seed=1234 + rank, device=dev,      # one independent sequence per GPU
                                kind="dimp" if cfg_name == "dimp50" else "prdimp")
    pool = make_pool(cfg, 4321 + rank, dev)
    K, Wm = args.steps, args.warmup
    stream = torch.cuda.Stream(device=dev)

    if args.profile_child:                                     # the leg rocprofv3 traces: eager frames, nothing else
        with torch.cuda.stream(stream):
            run_frames(st, pool, 0, Wm + K)
            stream.synchronize()
        return

    # ---- kernel-level legs first (rocprofv3 child process, then the event-pair periods of the two passes in this process):
    #      the frame timing that follows starts on a device that has just been busy, whatever --warmup says
    want_roof = not args.no_roofline and rank == 0 and world == 1
    stats, why = rocprof_kernel_stats(cfg_name) if want_roof else (None, "not requested")

    # ---- frame launcher: hipGraphs of G consecutive frames starting at frame --warmup, G = the timed steps themselves when
    #      they fit one memory cycle, else their common divisor with the memory size (whole graph replays; a graph is valid
    #      for one start slot, so one is captured per distinct start slot; every memory slot keeps being overwritten in
    #      turn); the warm-up frames as one more graph.  Eager launches when G < 5 / --no-graph.
    G = K if K <= n else math.gcd(K, n)
    use_graph = (not args.no_graph) and G >= 5
    graphs = {}
    with torch.cuda.stream(stream):
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

        period = None
        if want_roof:
            period = {"corr": event_period_us(st, stream, 0), "adj": event_period_us(st, stream, 1)}
        if use_graph and warm_graph is not None:
            warm_graph.replay()
        else:
            advance(0, Wm)
        stream.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        advance(Wm, K)
        stream.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        elapsed = time.perf_counter() - t0

        roof = roofline(cfg, cfg_name, n, period, stats, why) if want_roof else None

    # the only collective: the end-of-batch (frames, seconds) gather; whole-job rate = all frames / slowest rank
    total_frames, tmax, _ = sequences.gather_throughput(K, elapsed, device=dev)
    value = total_frames / tmax

    if rank == 0:
        solve_bytes = st.bytes_per_solve(NUM_ITER)
        if roof is not None:
            gbs = solve_bytes * (K / tmax) / 1e9
            roof["solve_level"] = {"algorithmic_bytes_per_frame": solve_bytes, "achieved_GBs": round(gbs, 1),
                                   "frac": round(gbs / HBM_PEAK_GBS, 4),
                                   "note": "2 feature reads per iteration x 5 iterations (SURVEY 8d) / measured frame time"}
        launch = (f"hipGraph replay, {G} frames per graph ({K // G} replay(s) in the timed region, {len(graphs)} start slot(s))" if use_graph
                  else "eager (18 launches per frame)")
        out = {
            "metric": "frames/sec DiMP-50 online track (288x288, 5 SD iters)" if cfg_name == "dimp50" else "frames/sec PrDiMP-50 online track (352x352, 5 SD iters)", "value": round(value, 2),
            "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": Wm,
            "ms_per_step": round(1e3 * tmax / K, 5), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("BASELINE configs[1]: DiMP-50 single sequence per GPU; per frame classify(1x512x18x18) "
                                    "+ arg-max + memory insert + DiMPSteepestDescentGN(5 it) over n=50x512x18x18, K=4")
                       if cfg_name == "dimp50" else
                       ("BASELINE configs[2] per-GPU workload: PrDiMP-50 single sequence per GPU; per frame "
                        "classify(1x512x22x22) + arg-max + memory insert + PrDiMPSteepestDescentNewton(5 it) over "
                        "n=50x512x22x22, K=4"),
                       "sequences_per_gpu": 1, "launch": launch, "parallelism": f"{world} independent sequences"},
            "roofline": roof,
        }
        if world == 1 and cfg_name == "dimp50" and not args.no_roofline:
            out["head_inclusive"] = head_inclusive(cfg, n, dev, stream)
        if world == 1 and cfg_name == "dimp50":
            if not args.no_other:
                out["other_workloads"] = workloads.all_other(dev)
            if not args.no_cpu_baseline:
                out["cpu_baseline"] = stock_baseline(cfg, n, "cpu", budget_s=12.0)
                one = stock_baseline(cfg, n, "cpu", budget_s=5.0, min_frames=3, threads=1)
                out["cpu_baseline"]["one_thread"] = {"value": one["value"], "unit": "frames/s", "cores": 1, "sample": one["sample"]}
                torch.set_num_threads(min(CPU_THREADS, os.cpu_count() or 1))
            if not args.no_gpu_baseline:
                out["gpu_stock_baseline"] = stock_baseline(cfg, n, dev, budget_s=4.0, min_frames=50)
            if not args.no_other:                                # last: its MIOpen find mode must not touch the baseline above
                try:
                    out["end_to_end"] = workloads.end_to_end(dev)
                except Exception as exc:                         # noqa: BLE001
                    out["end_to_end"] = {"error": f"{type(exc).__name__}: {exc}"[:300]}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
