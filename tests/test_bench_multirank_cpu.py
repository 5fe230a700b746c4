"""bench.py's multi-rank control flow end to end on a box without a GPU: the file is launched exactly as the
driver launches it -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P
bench.py --gpus N --steps K --warmup W` -- with PT_BENCH_DRYRUN=1 (gloo instead of RCCL, a stub sequence state; everything else is
the code the GPU run executes: rank wiring, build-on-rank-0 + barrier, warm-up, barrier-bracketed timed region, closing barrier,
the 16-byte gather, ONE JSON line on rank 0's stdout).  Mirrors the reference's one-sequence-per-worker pool
(pytracking/evaluation/running.py:198-218)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _launch(n, extra, via_launcher=True):
    env = dict(os.environ, PT_BENCH_DRYRUN="1", OMP_NUM_THREADS="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    tail = [os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "20", "--warmup", "5"] + extra
    if via_launcher:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port())] + tail
    else:
        cmd = [sys.executable] + tail                    # plain `python bench.py --gpus N`: re-executes itself under the launcher
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, f"stdout must carry ONE line, got {len(lines)}: {res.stdout[:500]}"
    return json.loads(lines[0])


@pytest.mark.parametrize("n,extra,via", [(2, [], True), (2, [], False), (8, ["--workload", "prdimp50"], True)])
def test_multirank_line(n, extra, via):
    d = _launch(n, extra, via)
    assert d["n_gpus"] == n and d["steps"] == 20 and d["warmup"] == 5 and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert d["data"].startswith("dry-run")
    assert len(d["per_rank"]) == n and [r["rank"] for r in d["per_rank"]] == list(range(n))
    assert all(r["frames"] == 20 and r["seconds"] > 0 for r in d["per_rank"])
    assert d["collective"]["ranks"] == n and d["collective"]["backend"] == "gloo"
    # whole-job value = all frames / slowest rank (the dry run's 20 steps take a few hundred microseconds and `seconds` is printed
    # with 6 decimals: the round-off alone is up to 0.5 %)
    tmax = max(r["seconds"] for r in d["per_rank"])
    assert abs(d["value"] - 20 * n / tmax) / d["value"] < 2e-2
    assert abs(d["ms_per_step"] - 1e3 * tmax / 20) < 1e-3
    assert d["bracket"]["rank0_seconds_incl_closing_barrier"] >= d["per_rank"][0]["seconds"]
    assert f"{n} independent sequences" == d["config"]["parallelism"]
    if extra:
        assert "PrDiMP-50" in d["metric"] and "configs[2]" in d["config"]["workload"]
    else:
        assert "DiMP-50" in d["metric"] and "configs[1]" in d["config"]["workload"]


def test_launcher_rank_count_mismatch_is_refused():
    env = dict(os.environ, PT_BENCH_DRYRUN="1", WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(_free_port()))
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], cwd=ROOT, env=env, capture_output=True,
                         text=True, timeout=120)
    assert res.returncode != 0 and "launcher started 2 ranks" in (res.stderr + res.stdout)
