"""bench.py -- frames/sec of the DiMP-50 online-tracking hot path on MI355X (BASELINE.json metric).

One "step" = one synthetic tracking frame of BASELINE.json configs[1]:
    classify the 1x512x18x18 test feature with the current 512x4x4 filter -> on-device arg-max -> overwrite one slot
    of the n=50 sample memory (features + box) -> DiMPSteepestDescentGN, 5 iterations, over the 50x512x18x18 memory.
Everything is resident in HBM before the timed region; frames are strictly dependent (frame t+1 classifies with the
filter frame t produced).  One process per GPU, one independent sequence per GPU (weak scaling, no data-path
collective); RCCL is used only to agree on the slowest rank's time.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line on rank 0 (see README / DESIGN.md for the fields).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from pytracking_amd import _lib, bench_frame, sequences, synth  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
NUM_ITER = 5
POOL = 50                  # distinct synthetic test features cycled through (one per memory slot)


def make_pool(cfg, seed, device):
    rng = np.random.default_rng(seed)
    return torch.from_numpy(synth.clf_features(rng, POOL, cfg["C"], cfg["H"], cfg["W"], cfg["K"])).to(device)


def run_frames(st, pool, first, count):
    n = st.n
    for f in range(first, first + count):
        st.step(pool[f % POOL], slot=f % n, num_iter=NUM_ITER)


def cpu_baseline(cfg, n, budget_s=12.0, min_frames=10):
    """Reference CPU path (torch-CPU port, oracle/frame_port.py) on this host, same workload, bounded sample."""
    from oracle.frame_port import TorchCpuTracker
    host = os.cpu_count() or 1
    pool = make_pool(cfg, 99, "cpu")
    # the oneDNN convs of this path stop scaling (and collapse when oversubscribed) well below a 2-socket
    # host's core count: calibrate the thread count on 2 frames each and keep the fastest
    best, threads = None, 1
    for th in sorted({t for t in (8, 16, 32, 64, host) if t <= host}):
        tr = TorchCpuTracker(cfg, n, seed=1234, threads=th)
        tr.step(pool[0], 0, NUM_ITER)
        t0 = time.perf_counter()
        tr.step(pool[1], 1, NUM_ITER)
        dt = time.perf_counter() - t0
        if best is None or dt < best:
            best, threads = dt, th
        if dt > 2.0:
            break
    tr = TorchCpuTracker(cfg, n, seed=1234, threads=threads)
    for f in range(2):
        tr.step(pool[f % POOL], f % n, NUM_ITER)
    t0 = time.perf_counter()
    frames = 0
    while frames < min_frames or time.perf_counter() - t0 < budget_s:
        tr.step(pool[(frames + 2) % POOL], (frames + 2) % n, NUM_ITER)
        frames += 1
        if frames >= 2000:
            break
    dt = time.perf_counter() - t0
    return {"value": frames / dt, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{frames} frames of the same workload in {dt:.1f}s (torch-CPU port of the reference path, fp32; "
                      f"best of 8/16/32/64/all threads on a {host}-core host)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--no-graph", action="store_true", help="launch every frame eagerly instead of replaying hipGraphs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--workload", default="dimp50", choices=("dimp50", "prdimp50"),
                    help="dimp50 = BASELINE configs[1] (the metric's configuration); prdimp50 = configs[2]'s per-GPU workload")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
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

    # ---- frame launcher: eager, or hipGraphs of one full memory cycle (n frames) --------------------------
    stream = torch.cuda.Stream(device=dev)
    graph = None
    with torch.cuda.stream(stream):
        run_frames(st, pool, 0, 2)                         # first-touch / code-object load outside everything
        stream.synchronize()
        if not args.no_graph:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=stream):
                run_frames(st, pool, 0, n)                 # slots 0..n-1, pool entries 0..n-1

        def advance(first, count):
            f, end = first, first + count
            while f < end:
                if graph is not None and f % n == 0 and end - f >= n:
                    graph.replay()
                    f += n
                else:
                    run_frames(st, pool, f, 1)
                    f += 1

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

        # ---- roofline leg: kprof more frames, eagerly, with HIP events around every feature-pass launch on the stream
        #      it is launched on.  An event pair costs a few microseconds of its own: the library brackets a one-wave
        #      kernel that spins for exactly 5.00 us next to every adjoint launch, and (that bracket - 5.00 us) is subtracted
        #      from the brackets around the passes (rocprofv3's kernel durations in profiles/ are the check).
        roof = None
        if not args.no_roofline and rank == 0:
            L = _lib.lib()
            prof = ctypes.c_void_p()
            per_frame = 2 * NUM_ITER + 2
            kprof = min(K, 200)
            _lib.check(L.pt_profile_create(ctypes.byref(prof), kprof * per_frame), "pt_profile_create")
            L.pt_profile_attach(prof)
            run_frames(st, pool, Wm + K, kprof)
            stream.synchronize()
            L.pt_profile_attach(None)
            feat_bytes = 4 * n * cfg["C"] * cfg["H"] * cfg["W"]       # one pass streams the n-sample memory once
            kern = {}
            for kid, name in ((0, "k_corr2"), (1, "k_adj2"), (2, "spin5us")):
                ms, cnt = ctypes.c_double(), ctypes.c_long()
                _lib.check(L.pt_profile_collect(prof, kid, ctypes.byref(ms), ctypes.byref(cnt)), "pt_profile_collect")
                kern[name] = (ms.value * 1e3 / max(cnt.value, 1), cnt.value)       # mean microseconds, launches
            L.pt_profile_destroy(prof)
            # the spin kernel's own duration as rocprofv3 sees it: 5.00 us of spinning + 0.58 us dispatch/drain of a
            # one-wave kernel (profiles/r01e_kernel_stats.csv: k_prof_spin avg 5581 ns)
            overhead = max(kern["spin5us"][0] - 5.58, 0.0)
            stats = {k: {"avg_launch_us": round(max(kern[k][0] - overhead, 1e-3), 3), "bracket_us": round(kern[k][0], 3),
                         "launches": kern[k][1],
                         "achieved_GBs": round(feat_bytes / max(kern[k][0] - overhead, 1e-3) / 1e3, 1)}
                     for k in ("k_corr2", "k_adj2")}
            dom = max(stats, key=lambda k: stats[k]["avg_launch_us"] * stats[k]["launches"])
            traffic, traffic_src = None, None
            pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")   # committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE pass
            if os.path.exists(pmc):
                rec = json.load(open(pmc)).get(cfg_name, {}).get(dom)
                if rec:
                    traffic, traffic_src = rec["hbm_bytes_per_launch"], rec["source"]
            roof = {"bound": "hbm", "kernel": dom, "achieved": stats[dom]["achieved_GBs"], "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(stats[dom]["achieved_GBs"] / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "traffic_source": traffic_src, "avg_launch_us": stats[dom]["avg_launch_us"],
                    "event_pair_overhead_us": round(overhead, 3), "launches": stats[dom]["launches"],
                    "algorithmic_bytes_per_launch": feat_bytes, "kernels": stats,
                    "solve_level": {"algorithmic_bytes_per_frame": st.bytes_per_solve(NUM_ITER), "achieved_GBs": None}}

    # the only collective: the end-of-batch (frames, seconds) gather; whole-job rate = all frames / slowest rank
    total_frames, tmax, _ = sequences.gather_throughput(K, elapsed, device=dev)
    value = total_frames / tmax

    if rank == 0:
        if roof is not None:
            roof["solve_level"]["achieved_GBs"] = round(st.bytes_per_solve(NUM_ITER) * (K / tmax) / 1e9, 1)
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
                       "sequences_per_gpu": 1, "launch": "eager" if graph is None else f"hipGraph of {n} frames",
                       "parallelism": f"{world} independent sequences"},
            "roofline": roof,
        }
        if not args.no_cpu_baseline and world == 1 and cfg_name == "dimp50":
            out["cpu_baseline"] = cpu_baseline(cfg, n)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
