"""bench.py --gpus N launches its own N ranks: the launcher, the process group and every N > 1 branch
of the three workloads (barriers, max-over-ranks clock, PredictionGather's asynchronous gather, the ragged gathers) driven
on CPU tensors over gloo with a stand-in for the GPU step (``--stub``), plus the loud failure when ranks outnumber devices.
The reference spawns its ranks itself too: scripts/run_pipeline.py:195-206."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")
for _p in (ROOT, os.path.join(ROOT, "open3d-ml_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)


def _run(argv, timeout=300):
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    return subprocess.run([sys.executable, BENCH] + argv, env=env, capture_output=True, text=True, timeout=timeout,
                          stdin=subprocess.DEVNULL)


def _line(r):
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    return json.loads(lines[0])


@pytest.mark.parametrize("workload", ["randlanet", "kpconv", "pointpillars"])
def test_self_launch_two_ranks_over_gloo(workload):
    out = _line(_run(["--gpus", "2", "--stub", "--steps", "3", "--warmup", "2", "--frames-per-step", "2", "--workload", workload]))
    assert out["stub"] is True and out["n_gpus"] == 2 and out["steps"] == 3
    assert out["gathered_ranks_checked"] == 2           # rank 0 verified BOTH ranks' predictions of the last step
    # ... and the check the REAL N > 1 run carries (round 6): every rank's label checksum recomputed on the gathered buffers
    if workload == "randlanet":
        assert out["gather_self_check"]["ranks_checked"] == 2 and out["gather_self_check"]["all_match"] is True
    seen = out["ranks_seen"]
    assert seen["world_size"] == 2 and len(seen["devices"]) == 2 and len(set(seen["devices"])) == 2


def test_self_launch_three_ranks_headline():
    out = _line(_run(["--gpus", "3", "--stub", "--steps", "2", "--warmup", "1", "--frames-per-step", "1"]))
    assert out["n_gpus"] == 3 and out["ranks_seen"]["world_size"] == 3 and out["self_launched"] is True


def test_self_launch_eight_ranks_are_pinned_to_disjoint_cpus():
    """The driver's 8-GPU launch shape: eight ranks over gloo, each pinned to its own CPU slice
    (ml3d.dist.bind_rank; no GPUs here, so the slices are an even split of the allowed CPUs), the map printed in ranks_seen."""
    out = _line(_run(["--gpus", "8", "--stub", "--steps", "2", "--warmup", "1", "--frames-per-step", "1"], timeout=600))
    assert out["n_gpus"] == 8 and out["ranks_seen"]["world_size"] == 8 and out["gathered_ranks_checked"] == 8
    assert out["gather_self_check"]["all_match"] is True and len(out["gather_self_check"]["per_rank"]) == 8
    cmap = out["ranks_seen"]["cpu_map"]
    assert len(cmap) == 8 and all(m["host_threads"] >= 1 and m["bound"] for m in cmap)
    from ml3d.dist import _parse_cpulist
    sets = [set(_parse_cpulist(m["cpus"])) for m in cmap]
    assert all(sets)
    if len(os.sched_getaffinity(0)) >= 8:              # enough cores: no two ranks share one
        assert sum(len(x) for x in sets) == len(set().union(*sets))


def test_single_rank_stub_needs_no_process_group():
    out = _line(_run(["--stub", "--steps", "2", "--warmup", "1", "--frames-per-step", "2"]))
    assert out["n_gpus"] == 1 and out["ranks_seen"]["world_size"] == 1 and out["self_launched"] is False


def test_gpus_must_match_world_size_under_torchrun():
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--stub"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "--gpus 2 but WORLD_SIZE 1" in r.stderr


def test_a_failing_rank_takes_the_job_down():
    # rank 1 of 2 dies at start-up (its --gpus / WORLD_SIZE check) -> the launcher must not hang on rank 0's rendezvous
    sys.path.insert(0, ROOT)
    import bench
    rc = bench.launch_ranks(2, ["--gpus", "5", "--stub"], stub=True)
    assert rc != 0


@pytest.mark.skipif(torch.cuda.is_available(), reason="needs a host WITHOUT a GPU")
def test_more_ranks_than_devices_fails_loudly_without_a_gpu():
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0 and "2 ranks, 0 devices" in r.stderr and not r.stdout.strip()


@pytest.mark.gpu
def test_more_ranks_than_devices_fails_loudly_on_the_gpu_box():
    n = torch.cuda.device_count()
    r = _run(["--gpus", str(n + 1), "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0 and ("%d ranks, %d device" % (n + 1, n)) in r.stderr and not r.stdout.strip()
