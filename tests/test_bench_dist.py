"""bench.py's N > 1 leg, end to end (the reference's counterpart is the partition loop of
SufrBuilder::write, sufr_builder.rs:875-906).

On the one-GPU box the leg runs in the two forms a single device allows:
  * `--gpus 2 --backend gloo --share-device`: two ranks started by bench.py ITSELF (no launcher on the command line), both on
    cuda:0, first-digit shards, the host form of the {first, last, count} exchange, the sharded `sufr create` into ONE file;
  * `SUFR_BENCH_FORCE_DIST=1 --gpus 1`: RCCL (backend nccl) initialised at world size 1, two shards per step, the device
    form of the exchange (all_gather_into_tensor on device tensors) feeding sufr_hip_stitch_device_u32 on the context's
    stream -- the composition a rank of an 8-GPU job runs, on the hardware we have."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
BENCH = str(ROOT / "bench.py")


def _run(args, env_extra=None, timeout=900):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "SUFR_BENCH_FORCE_DIST"):
        env.pop(k, None)
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, timeout=timeout, env=env, cwd=str(ROOT))
    return r


def _line(r):
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_gpus_n_without_enough_devices_says_how_to_rehearse():
    """No launcher, fewer GPUs than ranks and no --share-device: a message and exit code 2 instead of the old SystemExit
    that asked for torch.distributed.run (decided before anything touches a GPU: runs in the CPU container too)."""
    import torch
    if torch.cuda.device_count() >= 64:
        pytest.skip("a box with 64 GPUs")
    r = _run(["--gpus", "64", "--steps", "1", "--warmup", "0"])
    assert r.returncode == 2
    assert "--share-device" in r.stderr and "SUFR_BENCH_FORCE_DIST" in r.stderr


def test_share_device_with_rccl_is_refused_before_anything_starts():
    """N RCCL ranks on one device die with a duplicate-GPU error in every rank (advisor r5): the parent says what to add and
    exits 2 (decided before anything touches a GPU: runs in the CPU container too)."""
    r = _run(["--gpus", "2", "--share-device", "--steps", "1", "--warmup", "0"])
    assert r.returncode == 2 and "--backend gloo" in r.stderr


@pytest.mark.gpu
def test_bench_eight_ranks_end_to_end():
    """BASELINE config C5 at its real shape, rehearsed on the one GPU: `python bench.py --gpus 8 --backend gloo
    --share-device` -- eight ranks, eight first-digit shards, the eight-writer `sufr create` into ONE file -- at C. elegans size.
    Eight per-rank records whose shards add up to the N = 1 count; the file is sha256-equal to the single-GPU file."""
    common = ["--workload", "elegans", "--steps", "2", "--warmup", "1", "--placement-trials", "1", "--e2e-hash"]
    eight = _line(_run(["--gpus", "8", "--backend", "gloo", "--share-device"] + common, timeout=1500))
    one = _line(_run(["--gpus", "1", "--no-cpu-baseline", "--no-search"] + common))
    assert eight["n_gpus"] == 8 and eight["scaling"] == "strong"
    pr = eight["per_rank"]
    assert len(pr["device_ms"]) == 8 and sum(r["num_suffixes"] for r in pr["device_ms"]) == one["config"]["num_suffixes"]
    assert min(r["num_suffixes"] for r in pr["device_ms"]) > 0
    e8, e1 = eight["e2e_create"], one["e2e_create"]
    assert "error" not in e8 and "error" not in e1, (e8, e1)
    assert len(e8["shard_suffixes"]) == 8 and sum(e8["shard_suffixes"]) == one["config"]["num_suffixes"]
    assert e8["sufr_sha256"] == e1["sufr_sha256"] and len(e8["sufr_sha256"]) == 64


@pytest.mark.gpu
@pytest.mark.parametrize("workload", ["ecoli", "elegans"])
def test_bench_two_ranks_end_to_end(workload):
    """`python bench.py --gpus 2 --backend gloo --share-device`: bench.py spawns its two ranks; the JSON line says n_gpus 2,
    the shards add up to the N = 1 suffix count, and the sharded e2e_create file is byte-identical (sha256) to the file the
    single-GPU `sufr create` of the N = 1 run wrote."""
    common = ["--workload", workload, "--steps", "2", "--warmup", "1", "--placement-trials", "1", "--e2e-hash"]
    two = _line(_run(["--gpus", "2", "--backend", "gloo", "--share-device"] + common))
    one = _line(_run(["--gpus", "1", "--no-cpu-baseline", "--no-search"] + common))
    assert two["n_gpus"] == 2 and one["n_gpus"] == 1
    assert two["steps"] == 2 and two["warmup"] == 1 and two["scaling"] == "strong"
    assert two["config"]["num_suffixes"] == one["config"]["num_suffixes"] > 0
    assert two["config"]["text_len"] == one["config"]["text_len"]
    assert two["value"] > 0 and two["ms_per_step"] > 0
    # per-rank record: both ranks, the fixed cost named, the shared device flagged
    pr = two["per_rank"]
    assert len(pr["device_ms"]) == 2 and sum(r["num_suffixes"] for r in pr["device_ms"]) == one["config"]["num_suffixes"]
    assert all(t > 0 for t in pr["text_pass_ms"]) and "share_device" in pr
    e2, e1 = two["e2e_create"], one["e2e_create"]
    assert "error" not in e2 and "error" not in e1, (e2, e1)
    assert sum(e2["shard_suffixes"]) == one["config"]["num_suffixes"] and min(e2["shard_suffixes"]) > 0
    assert e2["sufr_bytes"] == e1["sufr_bytes"]
    assert e2["sufr_sha256"] == e1["sufr_sha256"] and len(e2["sufr_sha256"]) == 64


@pytest.mark.gpu
def test_bench_forced_dist_runs_the_rccl_step_on_one_gpu():
    """SUFR_BENCH_FORCE_DIST=1: nccl at world size 1; every step builds two shards, gathers the boundary triples over RCCL on
    device tensors and stitches on the context's stream; the stitched shards concatenate to the single build, element for
    element (checked inside bench.py on the arrays of the last timed step, outside the timed region)."""
    j = _line(_run(["--gpus", "1", "--workload", "elegans", "--steps", "3", "--warmup", "1", "--placement-trials", "1",
                    "--no-cpu-baseline", "--no-e2e"], env_extra={"SUFR_BENCH_FORCE_DIST": "1"}))
    f = j["forced_dist"]
    assert j["n_gpus"] == 1 and f["virtual_shards"] == 2 and "nccl" in f["backend"]
    assert f["equal_to_single_build"] is True
    assert sum(f["shard_suffixes"]) == j["config"]["num_suffixes"] and min(f["shard_suffixes"]) > 0
    assert f["stitched_first_lcp"] is not None and f["stitched_first_lcp"] < 100_000
    assert j["verified"] is None and j["roofline"]["traffic"] is None      # not the headline line: no borrowed counters
