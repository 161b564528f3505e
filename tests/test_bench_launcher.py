"""CPU: bench.py's multi-process path -- `python bench.py --gpus N` without a launcher starts its N ranks itself (self_launch ->
torch.distributed.run on 127.0.0.1), shards the image list, all-gathers per step and prints ONE JSON line on rank 0.  Driven here
with the gloo backend and --selftest-stub (the model step is a constant tensor; the line says so) -- the launcher, the sharding,
both all-gather payloads and the barrier / max-over-ranks timing are the real code."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT") and not k.startswith("I2R_")}
    env["OMP_NUM_THREADS"] = "1"
    return env


def test_gpus2_self_launch_gloo_stub():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--backend", "gloo",
                        "--selftest-stub", "--config", "hrt_192_p4_b4"], cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout  # ONE line, from rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["scaling"] == "weak"
    assert out["config"]["crops_per_gpu_step"] == 16 and "dp2" in out["config"]["parallelism"] and "key points" in out["config"]["parallelism"]
    assert out["gather_alt"]["payload"] == "heatmaps" and out["gather_alt"]["ms_per_step"] > 0
    assert "STUB" in out["data"] and out["value"] == 0.0  # a stub run can never be mistaken for a measurement


def test_gpus1_world1_collective_strong_gloo_stub():
    """--world1-collective: the N = 1 step with the collective in place (a ONE-rank process group: gather, barriers, max-over-ranks
    reduction) -- here on gloo with the stub step; tests/test_dist_gpu.py runs the real model over RCCL.  Strong scaling: the line
    carries `shards` (all 64 images on the one rank, 4 forwards of uneven crop counts per step, gathered together)."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--backend", "gloo",
                        "--selftest-stub", "--world1-collective", "--scaling", "strong"], cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    (line,) = [l for l in p.stdout.splitlines() if l.startswith("{")]
    out = json.loads(line)
    assert out["n_gpus"] == 1 and out["scaling"] == "strong" and "dp1" in out["config"]["parallelism"]
    sh = out["shards"]
    assert sh["images_total"] == 64 and sh["crops_per_rank"] == [sh["crops_total"]] and sh["forwards_per_rank_step"] == [4]
    assert out["gather_alt"]["payload"] == "heatmaps"
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--world1-collective", "--backend", "gloo", "--selftest-stub"],
                         cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=600)
    assert bad.returncode != 0 and "world1-collective" in (bad.stdout + bad.stderr)


def test_gpus2_without_launcher_reaches_process_group_init():
    """On a GPU-less box the real (nccl) path must get as far as creating the process group -- not exit at argument time."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU box: the real path is covered by the driver's --gpus runs")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                       cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=600)
    assert p.returncode != 0
    err = p.stdout + p.stderr
    assert "needs one process per GPU" not in err
    assert "torch.distributed.run" in err or "torch/distributed" in err or "ProcessGroupNCCL" in err or "NCCL" in err or "cuda" in err.lower(), err[-3000:]
    assert "init_process_group" in err or "ProcessGroupNCCL" in err or "set_device" in err or "No HIP GPUs" in err or "NCCL" in err, err[-3000:]


def test_op_model_covers_every_launch_kind():
    """bench.op_model gives every op kind of include/i2r_hip.h a kernel name (the roofline must be able to name ANY dominant kernel)"""
    sys.path.insert(0, ROOT)
    import bench
    from i2r_amd import cabi
    kinds = {v for k, v in vars(cabi).items() if k.startswith("OP_") and isinstance(v, int)} - set(cabi.SYNC_OPS)
    src = open(os.path.join(ROOT, "bench.py")).read()
    for name, v in vars(cabi).items():
        if name.startswith("OP_") and isinstance(v, int) and v in kinds:
            assert "cabi.%s" % name in src, name
    # two element-wise kinds on plain structs (no library call involved)
    st = cabi.LnArgs(0, 0, 0, 0, 1000, 78, 80, 1e-6, 1)
    name, flop, nbytes, pipe = bench.op_model(cabi.OP_LAYERNORM, st, "bf16")
    assert name == "layernorm_k" and pipe is None and nbytes == 1000 * 80 * 6
    st = cabi.HrtMlpArgs(*([0] * 10), 16, 64, 48, 78, 80, 320, 1e-6, 1)
    name, flop, nbytes, pipe = bench.op_model(cabi.OP_HRT_MLP, st, "bf16")
    assert name == "hrt_mlp_block_k" and pipe == "bf16" and abs(flop - 16 * 64 * 48 * (16 * 78 * 78 + 72 * 78)) < 1


def test_gpus2_strong_scaling_gloo_stub():
    """--scaling strong: a fixed ragged list of 64 images cut by dist.shard_bounds -> uneven crop counts per rank go through the padded
    all-gather on the timed path; the line reports the shards and their imbalance"""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--backend", "gloo",
                        "--selftest-stub", "--scaling", "strong"], cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    sys.path.insert(0, ROOT)
    import bench
    from i2r_amd import dist as i2r_dist
    total = sum(bench.STRONG_LENGTH)
    assert len(bench.STRONG_LENGTH) == 64 and all(1 <= n <= 6 for n in bench.STRONG_LENGTH)
    sh = out["shards"]
    assert out["scaling"] == "strong" and out["n_gpus"] == 2
    assert sh["images_total"] == 64 and sh["crops_total"] == total and sum(sh["crops_per_rank"]) == total
    assert sh["image_bounds"] == i2r_dist.shard_bounds(bench.STRONG_LENGTH, 2)
    assert sh["crops_per_rank"][0] != sh["crops_per_rank"][1] or total % 2 == 0  # (uneven unless the list happens to split exactly)
    assert 1.0 <= sh["imbalance_max_over_mean"] < 1.1  # at most one image (<= 6 crops) off an equal share of ~110 crops
    assert sh["gather_rows_padded_to"] == max(sh["crops_per_rank"])
    b = sh["image_bounds"]
    assert sh["forwards_per_rank_step"] == [-(-(b[r + 1] - b[r]) // 16) for r in range(2)]  # batches of <= 16 images (31 / 33 images: 2 / 3)
    assert out["config"]["crops_per_gpu_step"] == sh["crops_per_rank"][0]
    assert "STUB" in out["data"] and out["value"] == 0.0
