"""Data-parallel paths on the device: bench.py's launcher behaviour and RCCL initialisation, and 2-rank checks of the
cfg-3 gradient average and of the sharded kNN sweep (RCCL when the box has >= 2 GPUs, otherwise two processes on the
one GPU through gloo -- the code above the collective is the same)."""
import json
import os
import socket
import subprocess
import sys

import pytest

from helpers import ROOT

pytestmark = pytest.mark.gpu


def _port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _env():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["OMP_NUM_THREADS"] = "4"
    return env


def _torchrun(nproc, script, *args, timeout=900):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr",
           "127.0.0.1", "--master-port", str(_port()), script] + list(args)
    p = subprocess.run(cmd, env=_env(), cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    return p.stdout


def test_bench_refuses_more_gpus_than_visible():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64"], env=_env(), cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 3
    assert "only" in json.loads(p.stdout.strip().splitlines()[-1])["error"]


def test_bench_knn_leg_under_the_launcher_initialises_rccl():
    """What the driver does for N > 1, at N = 1: torch.distributed.run + backend "nccl" (= RCCL)."""
    out = _torchrun(1, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1", "--workload", "knn")
    line = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    assert {k: line["rccl"][k] for k in ("backend", "world_size", "allreduce_of_ones", "devices_shared")} == {
        "backend": "nccl", "world_size": 1, "allreduce_of_ones": 1.0, "devices_shared": False}
    assert line["n_gpus"] == 1 and line["unit"] == "queries/s" and line["value"] > 1e6


def test_bench_gpus_2_end_to_end_on_this_box():
    """`python bench.py --gpus 2` as the driver would run it on a 2-GPU node -- the self-launcher, one rank per GPU (two ranks
    sharing the one GPU through gloo when the box has a single device: LVC_BENCH_ALLOW_SHARED_GPU=1), rank-sharded synthetic
    images, per-rank CPU pinning, max-over-ranks timing, the two data-parallel legs (cfg-3 step with the gradient exchange and
    equal parameters on every rank, sharded kNN sweep gathered to rank 0) and ONE merged JSON line from rank 0."""
    import torch

    env = _env()
    env.pop("OMP_NUM_THREADS", None)
    if torch.cuda.device_count() < 2:
        env["LVC_BENCH_ALLOW_SHARED_GPU"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                        "--no-live-pmc", "--pipeline-depth", "1"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines                      # rank 0 alone prints
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["config"]["global_batch"] == 16
    assert line["rccl"]["world_size"] == 2 and line["rccl"]["allreduce_of_ones"] == 2.0
    assert line["rccl"]["backend"] == ("nccl" if torch.cuda.device_count() >= 2 else "gloo")
    assert len(line["per_rank"]["img_per_s"]) == 2 and line["value"] > 0
    assert line["config"]["detections_per_image"] == [100] * 8
    legs = line["dp_legs"]
    assert legs["train_cfg3"]["parameters_identical_across_ranks"] is True and legs["train_cfg3"]["global_batch"] == 16
    assert legs["knn"]["top10_rows_identical_to_oracle"] if "top10_rows_identical_to_oracle" in legs["knn"] else legs["knn"]["queries_per_s"] > 0


def test_bench_gpus_8_rehearsal_on_this_box():
    """The real world size (BASELINE's whole-node configuration) rehearsed on whatever this box has: `python bench.py --gpus 8`
    through the self-launcher -- eight processes, each with its own stream-K workspaces and persistent conv workers, sharing the
    one GPU's CUs when the box has a single device (gloo instead of RCCL then) -- both data-parallel legs and ONE merged JSON line
    with eight per-rank entries.  Exercises what a 1-GPU box can of the 8-rank path: launcher, rank-sharded inputs (InferenceSampler
    blocks 0..7, reference detectron2/data/samplers/distributed_sampler.py:191-194), per-rank CPU slices, max-over-ranks timing,
    the gradient exchange over eight ranks with identical parameters afterwards, the 8-way sharded kNN sweep, and spinning stream-K
    workers of eight processes resident together (no time-out bit, equal detections on every rank)."""
    import torch

    env = _env()
    env.pop("OMP_NUM_THREADS", None)
    if torch.cuda.device_count() < 8:
        env["LVC_BENCH_ALLOW_SHARED_GPU"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                        "--no-live-pmc", "--pipeline-depth", "1"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=1700)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["scaling"] == "weak" and line["config"]["global_batch"] == 64
    assert line["rccl"]["world_size"] == 8 and line["rccl"]["allreduce_of_ones"] == 8.0
    assert line["rccl"]["backend"] == ("nccl" if torch.cuda.device_count() >= 8 else "gloo")
    assert len(line["per_rank"]["img_per_s"]) == 8 and len(line["per_rank"]["seconds"]) == 8 and line["value"] > 0
    assert line["config"]["detections_per_image"] == [100] * 8
    legs = line["dp_legs"]
    assert legs["train_cfg3"]["parameters_identical_across_ranks"] is True and legs["train_cfg3"]["global_batch"] == 64
    assert legs["knn"]["queries_per_s"] > 0
    # the line explains itself: who ran where, what the host spent per rank, how much of the gradient exchange was hidden
    dm = line["rccl"]["device_map"]
    assert [d["rank"] for d in dm] == list(range(8)) and all(d["pid"] > 0 and d["device_name"] for d in dm) and len({d["pid"] for d in dm}) == 8
    assert len(line["per_rank"]["host_enqueue_us_per_step"]) == 8 and all(h > 0 for h in line["per_rank"]["host_enqueue_us_per_step"])
    ex = legs["train_cfg3"]["gradient_exchange"]["per_rank"]
    assert len(ex) == 8 and all(set(e) == {"exposed_ms", "alone_ms", "hidden_ms"} and e["alone_ms"] > 0 for e in ex)
    assert len(legs["train_cfg3"]["per_rank_img_per_s"]) == 8


def _worker(mode):
    out = _torchrun(2, os.path.join(ROOT, "tests", "_dp_worker.py"), mode)
    line = [l for l in out.splitlines() if l.startswith("DP_WORKER_RESULT ")][-1]
    return json.loads(line[len("DP_WORKER_RESULT "):])


def test_two_rank_gradient_average_equals_the_concatenated_batch():
    r = _worker("train")
    print(r)
    assert r["world"] == 2 and r["bytes"] == 103525 * 4
    # fp32 atomics in the weight-gradient kernel: summation order differs between the two evaluations
    assert max(r["rel_err_vs_concatenated_batch"]) <= 2e-4, r
    assert r["buckets_all_reduce"] <= 2e-4 and r["buckets_rs_ag"] <= 2e-4, r


def test_gradient_buckets_reduce_scatter_all_gather_over_rccl_world_1():
    """RCCL's reduce_scatter_tensor / all_gather_into_tensor path of GradientBuckets (mode "rs_ag"), at the world size this
    box has (1): the collectives run through RCCL and leave the gradients unchanged."""
    code = (
        "import os, sys, torch, torch.distributed as dist\n"
        "sys.path.insert(0, %r)\n"
        "from lvc_amd import distributed as D\n"
        "torch.cuda.set_device(0); dist.init_process_group('nccl', device_id=torch.device('cuda', 0))\n"
        "ps = [torch.nn.Parameter(torch.randn(n, device='cuda')) for n in (1000, 33, 70001)]\n"
        "b = D.GradientBuckets(ps, bucket_bytes=1 << 12, mode='rs_ag')\n"
        "loss = sum((p * p).sum() for p in ps); loss.backward(); n = b.finish()\n"
        "ok = all(torch.equal(p.grad, 2 * p.detach()) for p in ps)\n"
        "print('RSAG', ok, n, len(b.buckets), dist.get_backend()); dist.destroy_process_group()\n" % ROOT)
    path = os.path.join(ROOT, "gpurun_out", "_rsag_worker.py")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    open(path, "w").write(code)
    out = _torchrun(1, path)
    line = [l for l in out.splitlines() if l.startswith("RSAG")][-1].split()
    assert line[1] == "True" and int(line[2]) == (1000 + 33 + 70001) * 4 and int(line[3]) >= 2 and line[4] == "nccl", line


def test_two_rank_sharded_knn_equals_the_single_process_sweep():
    r = _worker("knn")
    print(r)
    assert r["world"] == 2 and r["rows"] == 4096 and r["top_equal"] and r["keep_equal"]
