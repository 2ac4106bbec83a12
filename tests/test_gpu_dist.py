"""The RCCL path on one rank (`-m gpu`): what the driver's 2/4/8-GPU bench uses, exercised on the 1-GPU box --
process-group initialisation with backend "nccl" (= RCCL on ROCm), the fused arg-min selection + all_reduce of the
[world,18] table, the multi-object table merge, and bench.py end to end with DDX_FORCE_DIST=1.  Run in subprocesses so
that the test process itself never holds a process group."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_SCRIPT = r"""
import os, sys, torch, numpy as np
sys.path.insert(0, %(root)r)
import torch.distributed as dist
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%(port)d", rank=0, world_size=1, device_id=dev)
from diffdope_amd import dist as ddist
g = torch.Generator(device="cpu").manual_seed(0)
rows = torch.rand(4, 64, generator=g).cuda()
rows[:, 40] = rows[:, 3] = rows.min(1).values - 0.5      # tie: index 3 wins
mtx = torch.rand(64, 16, generator=g).cuda()
ref = ddist.global_argmin(rows[[0, 2]].mean(0), mtx.reshape(64, 4, 4), lo=128)
got = ddist.global_argmin_fused(rows, 0b0101, mtx, lo=128)
assert got[0] == ref[0] == 131, (got[0], ref[0])
assert abs(got[1] - ref[1]) < 1e-6 and torch.equal(got[2].cpu(), ref[2].cpu())
tab = torch.rand(5, 18, generator=g).cuda()
merged = ddist.merge_object_tables(tab.clone())
torch.cuda.synchronize()
assert torch.equal(merged, tab)                           # one rank owns every row: SUM over one rank = identity
assert dist.get_backend() == "nccl"
dist.barrier()
dist.destroy_process_group()
print("RCCL_ONE_RANK_OK")
"""


def _free_port():
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_rccl_one_rank_argmin_and_object_table():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", _SCRIPT % dict(root=ROOT, port=_free_port())], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "RCCL_ONE_RANK_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_bench_runs_through_the_rccl_path_on_one_rank():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", DDX_FORCE_DIST="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "10", "--warmup", "3", "--no-cpu-baseline", "--no-extras"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["config"]["global_hypotheses"] == 64
    assert line["final_pose"]["argmin_global_index"] in range(64)


def test_bench_with_two_ranks_sharing_the_gpu():
    """bench.py at world size 2, launched the way the driver launches it (torch.distributed.run, two ranks).  RCCL refuses two
    ranks on one device, so the ranks share GPU 0 over gloo (DDX_BENCH_SHARE_GPU): every line of the N > 1 path runs -- shard
    offsets, global batch 128 in the batch mean, barrier-bracketed window, the [2,18] table all_reduce, max over ranks."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", DDX_BENCH_SHARE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "8", "--warmup", "3", "--no-cpu-baseline", "--no-extras"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1  # rank 0 only
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["config"]["global_hypotheses"] == 128 and line["config"]["hypotheses_per_gpu"] == 64
    assert line["value"] > 0 and abs(line["value"] - 2 * 1000.0 / line["ms_per_step"]) < 1e-6 * line["value"]  # whole-job rate = 2 ranks x iterations/s
    assert line["final_pose"]["argmin_global_index"] in range(128)


def test_config4_fixed_job_two_ranks_sharing_the_gpu_matches_one_process():
    """BASELINE configs[3] in the form the 8-GPU node runs it (--config cfg4 --global-batch G: a FIXED job sharded by
    dist.shard_range), at world size 2 on the one GPU (gloo, DDX_BENCH_SHARE_GPU) against the same 128-hypothesis job in one
    process: strong scaling reported, 64 hypotheses per rank, the same arg-min hypothesis and loss."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("DDX_BENCH_SHARE_GPU", None)
    common = ["--config", "cfg4", "--steps", "8", "--warmup", "3", "--global-batch", "128", "--no-cpu-baseline", "--no-extras", "--no-convergence"]
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + common, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert one.returncode == 0, one.stdout[-2000:] + one.stderr[-4000:]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2"] + common
    two = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(env, DDX_BENCH_SHARE_GPU="1"), cwd=ROOT)
    assert two.returncode == 0, two.stdout[-2000:] + two.stderr[-4000:]
    a = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    b = json.loads([l for l in two.stdout.splitlines() if l.startswith("{")][-1])
    assert a["scaling"] == b["scaling"] == "strong" and b["n_gpus"] == 2 and b["dist"]["world_size"] == 2
    assert a["config"]["global_hypotheses"] == b["config"]["global_hypotheses"] == 128
    assert a["config"]["hypotheses_per_gpu"] == 128 and b["config"]["hypotheses_per_gpu"] == 64
    assert a["final_pose"]["argmin_global_index"] == b["final_pose"]["argmin_global_index"]
    assert abs(a["final_pose"]["argmin_loss"] - b["final_pose"]["argmin_loss"]) <= 1e-3 * abs(a["final_pose"]["argmin_loss"])


def test_config4_as_the_node_runs_it_eight_ranks_sharing_the_gpu_matches_one_process():
    """Round 5 (the last thing that can be checked before a node exists): BASELINE configs[3] exactly as the 8-GPU node runs it --
    `bench.py --gpus 8 --config cfg4 --global-batch 512 --steps 20 --warmup 5` under torch.distributed.run with EIGHT ranks -- on
    the one GPU (gloo, DDX_BENCH_SHARE_GPU) against the same 512-hypothesis job in one process: eight distinct contiguous shard
    ranges of 64 in the line, world size 8, strong scaling, the same arg-min hypothesis / loss / pose error, and every rank's
    two-stream probe (eight processes share the GPU's hardware queues) came back with an answer instead of a hang."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("DDX_BENCH_SHARE_GPU", None)
    common = ["--config", "cfg4", "--steps", "20", "--warmup", "5", "--global-batch", "512", "--no-cpu-baseline", "--no-extras", "--no-convergence"]
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + common, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert one.returncode == 0, one.stdout[-2000:] + one.stderr[-4000:]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "8"] + common
    many = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=dict(env, DDX_BENCH_SHARE_GPU="1"), cwd=ROOT)
    assert many.returncode == 0, many.stdout[-2000:] + many.stderr[-4000:]
    lines = [l for l in many.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1  # rank 0 only
    a = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    b = json.loads(lines[0])
    assert b["n_gpus"] == 8 and b["dist"]["world_size"] == 8 and b["scaling"] == a["scaling"] == "strong"
    assert b["dist"]["shard_ranges"] == [[64 * r, 64 * r + 64] for r in range(8)]
    assert b["config"]["global_hypotheses"] == a["config"]["global_hypotheses"] == 512 and b["config"]["hypotheses_per_gpu"] == 64
    assert len(b["dist"]["two_chains"]) == 8 and all(x in (-1, 0, 1) for x in b["dist"]["two_chains"])
    assert a["final_pose"]["argmin_global_index"] == b["final_pose"]["argmin_global_index"]
    assert abs(a["final_pose"]["argmin_loss"] - b["final_pose"]["argmin_loss"]) <= 1e-3 * abs(a["final_pose"]["argmin_loss"])
    for k in ("rot_err_rad_best", "trans_err_m_best"):
        assert abs(a["final_pose"][k] - b["final_pose"][k]) <= 1e-3 * max(abs(a["final_pose"][k]), 1e-6) + 1e-7
    print("eight ranks on one GPU:", b["value"], "it/s of the 512-hypothesis job; one process:", a["value"], "| two-chain probe per rank:", b["dist"]["two_chains"])


def test_multi_object_frame_with_two_ranks_sharing_the_gpu():
    """examples/run_bop_scene.py (bop.refine_frame: objects sharded over ranks, one all_reduce of the object table) with two ranks on
    the one GPU over gloo: the same poses as the single-process run, every object's owner reported."""
    ex = os.path.join(ROOT, "examples", "run_bop_scene.py")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    one = subprocess.run([sys.executable, ex], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert one.returncode == 0, one.stdout[-2000:] + one.stderr[-4000:]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), ex]
    two = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(env, DDX_BENCH_SHARE_GPU="1"), cwd=ROOT)
    assert two.returncode == 0, two.stdout[-2000:] + two.stderr[-4000:]
    a = [l.split("(owner")[0] for l in one.stdout.splitlines() if l.startswith("object ")]
    b = [l.split("(owner")[0] for l in two.stdout.splitlines() if l.startswith("object ")]
    assert len(a) == 3 and a == b  # same arg-min hypothesis, loss and pose errors, digit for digit
    owners = [l.split("(owner rank ")[1].rstrip(")") for l in two.stdout.splitlines() if l.startswith("object ")]
    assert owners == ["0", "1", "0"]


def test_bench_two_ranks_over_rccl_matches_the_single_process_job():
    """First contact with RCCL at world size 2 (needs two GPUs: skipped on the 1-GPU test box, runs on the 8-GPU node): bench.py --gpus 2
    as the driver launches it -- one rank per device, backend nccl (= RCCL), no DDX_BENCH_SHARE_GPU -- on a FIXED job of 128 hypotheses
    (--global-batch 128: 64 per rank) against the same job in one process on one GPU: same arg-min hypothesis, same loss to fp32
    rounding of the differently grouped gradient sums, and the JSON line says which backend / devices ran."""
    import torch

    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL refuses two ranks on one device)")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("DDX_BENCH_SHARE_GPU", None)
    common = ["--steps", "8", "--warmup", "3", "--global-batch", "128", "--no-cpu-baseline", "--no-extras", "--no-convergence"]
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + common, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert one.returncode == 0, one.stdout[-2000:] + one.stderr[-4000:]
    two = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + common, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert two.returncode == 0, two.stdout[-2000:] + two.stderr[-4000:]
    a = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    b = json.loads([l for l in two.stdout.splitlines() if l.startswith("{")][-1])
    assert b["n_gpus"] == 2 and b["scaling"] == "strong" and b["dist"]["backend"] == "nccl" and b["dist"]["world_size"] == 2
    assert sorted(b["dist"]["device_ids"]) == [0, 1]
    assert a["config"]["global_hypotheses"] == b["config"]["global_hypotheses"] == 128 and b["config"]["hypotheses_per_gpu"] == 64
    assert a["final_pose"]["argmin_global_index"] == b["final_pose"]["argmin_global_index"]
    assert abs(a["final_pose"]["argmin_loss"] - b["final_pose"]["argmin_loss"]) <= 1e-3 * abs(a["final_pose"]["argmin_loss"])
