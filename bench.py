#!/usr/bin/env python
"""bench.py -- img/s of the R50-FPN GeneralizedRCNN inference forward on synthetic 800x1333 batches.

  python bench.py --gpus N --steps K --warmup W [--workload infer|train|knn]

N > 1 without a launcher re-executes itself under `python -m torch.distributed.run --nproc-per-node N` (one rank per
GPU over RCCL); launched BY torch.distributed.run (RANK / WORLD_SIZE in the environment) it runs as that rank.

Default workload `infer` (BASELINE.json configs[1]): one "step" = one call of the drop-in surface, `GeneralizedRCNN.forward(
batched_inputs) -> list[{"instances": Instances}]` (reference rcnn.py:100-125), on one batch of 8 synthetic 3x800x1333 images per GPU,
including the step's one device->host read; inputs are resident in HBM before the timed region; weights are the conditioned
random-init R50-FPN (lvc_amd/utils/synthetic.py).  Images shard data-parallel across ranks with no data-path collective
(reference InferenceSampler semantics), so scaling is "weak" and value = all images of all ranks / max-over-ranks time.
Rank 0 prints ONE JSON line carrying
  roofline            the dominant conv kernel, HIP events around its launches inside the timed region; plus `backbone_mfma_busy`
                      (matrix-pipe occupancy of the whole backbone, duration-weighted) and `step_traffic` (fabric bytes of the whole
                      step) from live rocprofv3 --pmc passes, against `step_algorithmic_bytes`
  bandwidth_kernels   ROIAlign / batched NMS / kNN top-k: event-timed ms, algorithmic bytes, GB/s and fraction of 8 TB/s
  knn                 BASELINE configs[3] on this GPU: 120k x 2400 x 1024 cosine sweep (ms, TF/s, GB/s, oracle agreement)
  value_inference_batched the same K steps through `inference_batched` (device tensors out, no host read in the loop)
  pipelined           the same K steps with two batches in flight on two HIP streams
  cpu_baseline        the CPU oracle timed on this host (rank 0, N = 1 only), 5 warm-up + 20 timed images
  rccl / per_rank     (N > 1 or under a launcher) the RCCL world and every rank's own rate
  dp_legs             (N > 1) short RCCL legs: cfg-3 fine-tune steps with the gradient all-reduce, sharded kNN sweep
  train               (N = 1) cfg-3 fine-tune step (bs 8), cfg-5 R101 box-corrector step (bs 2) with ms forward / backward /
                      optimizer, and R101-FPN inference img/s
  timed_batch_parity  the timed batch's detections against the CPU oracle's; bars = 2 x the oracle's own fp32-vs-fp64 noise on the
                      batch (oracle/noise.py); the run exits non-zero when a bar is missed
`--workload train` / `--workload knn` time those two data-parallel legs as the main metric instead.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16X3_TFLOPS = 2500.0 / 6  # dense bf16 MFMA peak / 6 bf16 MFMAs per fp32-accurate product
PEAK_F16X2_TFLOPS = 2500.0 / 3   # dense fp16 MFMA peak / 3 fp16 MFMAs per fp32-accurate product (two-way fp16 split)
PEAK_HBM_GBPS = 8000.0           # same guide: HBM3E 8 TB/s spec (6.3 TB/s measured achievable)
BATCH_PER_GPU = 8
KNN_Q, KNN_S, KNN_D = 120000, 2400, 1024   # BASELINE configs[3] / SURVEY 8(d) cfg 4


def _self_launch(args):
    """`python bench.py --gpus N` with no launcher: become the launcher (torch.distributed.run, one rank per GPU)."""
    import torch

    have = torch.cuda.device_count()
    if have < args.gpus and os.environ.get("LVC_BENCH_ALLOW_SHARED_GPU") != "1":
        print(json.dumps({"error": "bench.py --gpus %d: only %d GPU(s) visible" % (args.gpus, have), "n_gpus": args.gpus}))
        sys.exit(3)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL needs it on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd, env=env))


def _pin_rank_to_cpus(local_rank, world):
    """One slice of the host's CPUs per rank (contiguous: a slice stays on one socket / CCD group), and a thread pool sized
    for it: eight ranks' Python launch loops and torch CPU pools otherwise migrate over -- and oversubscribe -- the same
    cores.  Returns a description for the bench line, or None when the platform has no affinity call."""
    try:
        cpus = sorted(os.sched_getaffinity(0))
    except AttributeError:
        return None
    per = max(1, len(cpus) // max(1, world))
    mine = cpus[local_rank * per: (local_rank + 1) * per] or cpus
    try:
        os.sched_setaffinity(0, mine)
    except OSError:
        return None
    threads = max(1, min(8, len(mine)))
    os.environ["OMP_NUM_THREADS"] = str(threads)
    try:
        import torch

        torch.set_num_threads(threads)
    except Exception:
        pass
    return {"cpus": "%d-%d" % (mine[0], mine[-1]), "count": len(mine), "threads": threads}


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _event_ms(fn, iters, warm=2):
    """Average HIP-event time of fn() on torch's current stream (the stream every lvc_amd launch goes to)."""
    import torch

    for _ in range(warm):
        fn()
    e = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in e:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    return sum(a.elapsed_time(b) for a, b in e) / iters


class Ctx:
    pass


def _setup():
    import torch
    import torch.distributed as dist

    c = Ctx()
    c.rank = int(os.environ.get("RANK", "0"))
    c.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    c.world = int(os.environ.get("WORLD_SIZE", "1"))
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    ndev = torch.cuda.device_count()
    # LVC_BENCH_ALLOW_SHARED_GPU=1 (tests on a one-GPU box): ranks share devices round-robin and talk through gloo -- RCCL refuses
    # two ranks on one device.  Everything else (launcher, rank-sharded inputs, legs, JSON merge) is the N-GPU path.
    shared = ndev < c.world and os.environ.get("LVC_BENCH_ALLOW_SHARED_GPU") == "1"
    c.local_rank = c.local_rank % ndev if shared else c.local_rank
    torch.cuda.set_device(c.local_rank)
    c.dev = torch.device("cuda", c.local_rank)
    c.use_dist = c.world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)
    c.rccl = None
    if c.use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        pin = _pin_rank_to_cpus(int(os.environ.get("LOCAL_RANK", "0")), c.world) if c.world > 1 else None
        if shared:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=c.dev)  # "nccl" is RCCL on ROCm
        one = torch.ones(1, device=c.dev)
        dist.all_reduce(one)
        c.rccl = {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                  "allreduce_of_ones": float(one.item()), "devices_shared": bool(shared), "cpu_affinity_rank0": pin}
        assert dist.get_world_size() == c.world and float(one.item()) == c.world
        # who runs where (the first real N > 1 run explains itself): one entry per rank
        import socket

        props = torch.cuda.get_device_properties(c.dev)
        mine = {"rank": c.rank, "local_rank": int(os.environ.get("LOCAL_RANK", "0")), "device": c.local_rank, "device_name": props.name,
                "pci_bus_id": getattr(props, "pci_bus_id", None), "uuid": str(getattr(props, "uuid", "")), "host": socket.gethostname(), "pid": os.getpid(),
                "cpus": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None,
                "visible_devices": os.environ.get("HIP_VISIBLE_DEVICES", os.environ.get("ROCR_VISIBLE_DEVICES")), "devices_on_node": ndev}
        every = [None] * c.world
        dist.all_gather_object(every, mine)
        c.rccl["device_map"] = every
    return c


def _gather_obj(c, obj):
    """[obj of rank 0, ..] on every rank."""
    if not c.use_dist:
        return [obj]
    import torch.distributed as dist

    every = [None] * c.world
    dist.all_gather_object(every, obj)
    return every


def _barrier(c):
    import torch
    import torch.distributed as dist

    torch.cuda.synchronize()
    if c.use_dist:
        dist.barrier()
    torch.cuda.synchronize()


def _max_and_all(c, dt):
    """(max over ranks, list of every rank's value)."""
    import torch
    import torch.distributed as dist

    t = torch.tensor([dt], device=c.dev, dtype=torch.float64)
    if not c.use_dist:
        return dt, [dt]
    every = [torch.zeros_like(t) for _ in range(c.world)]
    dist.all_gather(every, t)
    vals = [float(x.item()) for x in every]
    return max(vals), vals


# ----------------------------------------------------------------------------------------------- kNN (cfg 4)
def _knn_inputs(c, rows, seed, classes=None, centers=None, spread=0.0, D=None):
    """Synthetic descriptors.  classes None: N(0, 1) rows (SURVEY 8(d) cfg 4's inputs).  `classes` given: class-structured rows
    centers[class] + spread * noise + 0.3 (what a descriptor network produces for objects of 80 classes: the premise of a
    kNN vote)."""
    import torch

    g = torch.Generator(device=c.dev).manual_seed(seed)
    x = torch.randn(rows, D or KNN_D, device=c.dev, generator=g)
    if classes is None:
        return x
    return centers[classes] + spread * x + 0.3


KNN_INPUTS = None


def knn_leg(c, steps, warmup, sample_check=2048):
    """BASELINE configs[3]: Q = 120 000 queries (sharded over the ranks: strong scaling), S = 2400 shots of 80 classes,
    D = 1024, cosine, k = 10.  Every rank contributes S / world shots to ONE RCCL all-gather, sweeps its own queries,
    results are gathered to rank 0 (reference tools/run_nearest_neighbours.py:301-325).
    Headline inputs: `randn(120000, 1024)` queries / `randn(2400, 1024)` shots, as SURVEY 8(d) defines cfg 4.  Timed next to
    them: class-structured descriptors (80 class directions, 30 shots each, queries of a random class, 30 % of the detector
    labels wrong) -- the two-stage sweep re-evaluates a neighbour in fp32 only where a shot of ANOTHER class is within the
    pre-filter's error margin, which unstructured rows make the common case -- and D = 384 (the ViT-S/8 descriptor width of
    tools/run_nearest_neighbours.py:292-293)."""
    import torch

    from lvc_amd import distributed as D
    from lvc_amd import label_verification as LV
    from lvc_amd.label_verification import knn_sweep, knn_sweep_distributed

    rng = D.shard_range(KNN_Q, c.rank, c.world)
    srng = D.shard_range(KNN_S, c.rank, c.world)
    classes_all = torch.arange(80).repeat_interleave(30)
    cls = classes_all[srng.start: srng.stop].to(c.dev)
    gc = torch.Generator(device=c.dev).manual_seed(5)
    centers = torch.randn(80, KNN_D, device=c.dev, generator=gc)      # the same on every rank
    gq = torch.Generator(device=c.dev).manual_seed(200 + c.rank)
    qcls = torch.randint(0, 80, (len(rng),), device=c.dev, generator=gq)
    det = torch.where(torch.rand(len(rng), device=c.dev, generator=gq) < 0.7, qcls, torch.randint(0, 80, (len(rng),), device=c.dev, generator=gq))
    inputs = {"randn": (_knn_inputs(c, len(srng), 7 + c.rank), _knn_inputs(c, len(rng), 100 + c.rank)),
              "structured": (_knn_inputs(c, len(srng), 7 + c.rank, cls, centers, 2.0), _knn_inputs(c, len(rng), 100 + c.rank, qcls, centers, 2.5))}

    def timed(shots, q, n):
        def step():
            if c.use_dist:
                return knn_sweep_distributed(cls, shots, q, det, 10, True)
            return knn_sweep(cls, shots, q, det, 10, True)

        for _ in range(max(1, warmup)):
            res = step()
        _barrier(c)
        t0 = time.perf_counter()
        for _ in range(n):
            res = step()
        _barrier(c)
        dt, _ = _max_and_all(c, time.perf_counter() - t0)
        return dt / n, res

    only = KNN_INPUTS                                # --knn-inputs: one kind of input per profiler trace ("randn" | "structured")
    per, (top, keep) = timed(*inputs[only or "randn"], steps)
    if only:
        per_s, top_s, keep_s = per, top, keep
    else:
        per_s, (top_s, keep_s) = timed(*inputs["structured"], steps)
    path = "two-stage (fp16 pre-filter GEMM + exact fp32 verification)" if LV.KNN_TWO_STAGE else "single-stage (fp32-accurate f16x2 GEMM + top-k)"
    fl = 2.0 * KNN_Q * KNN_S * KNN_D
    out = {"workload": "kNN label verification: Q=%d (sharded %d/rank) x S=%d x D=%d, cosine, top-10 + vote" % (KNN_Q, len(rng), KNN_S, KNN_D),
           "path": path, "inputs": "randn(Q, D) queries / randn(S, D) shots (SURVEY 8(d) cfg 4)" if not only else only,
           "ms_per_sweep": round(per * 1e3, 3), "queries_per_s": round(KNN_Q / per),
           "algorithmic_tflops": round(fl / per / 1e12, 1),
           "algorithmic_bytes": KNN_Q * KNN_D * 4 + KNN_S * KNN_D * 4 + KNN_Q * 10 * 8,
           "kept_fraction": round(float(keep.float().mean()), 4) if keep is not None else None}   # gathered to rank 0 only
    out["algorithmic_GBps"] = round(out["algorithmic_bytes"] / per / 1e9, 1)
    out["frac_of_hbm_peak"] = round(out["algorithmic_bytes"] / per / 1e9 / PEAK_HBM_GBPS, 4)
    # one fp16 MFMA per product in the pre-filter (dense fp16 peak); the single-stage path needs three (f16x2 peak)
    peak = PEAK_F16X2_TFLOPS * 3 if LV.KNN_TWO_STAGE else PEAK_F16X2_TFLOPS
    out["frac_of_mfma_peak"] = round(fl / per / 1e12 / peak, 4)
    out["mfma_peak_tflops"] = round(peak, 1)
    out["structured"] = {"inputs": "class-structured synthetic descriptors (80 classes x 30 shots, 30 % wrong detector labels)",
                         "ms_per_sweep": round(per_s * 1e3, 3), "queries_per_s": round(KNN_Q / per_s),
                         "kept_fraction": round(float(keep_s.float().mean()), 4) if keep_s is not None else None}
    if not only and not c.use_dist:
        # D = 384: the descriptor width of the ViT-S/8 the reference verifies with
        sh3, q3 = _knn_inputs(c, KNN_S, 17, D=384), _knn_inputs(c, KNN_Q, 117, D=384)
        for _ in range(2):
            knn_sweep(cls, sh3, q3, det, 10, True)
        _barrier(c)
        t0 = time.perf_counter()
        for _ in range(steps):
            knn_sweep(cls, sh3, q3, det, 10, True)
        _barrier(c)
        p3 = (time.perf_counter() - t0) / steps
        out["d384"] = {"workload": "Q=%d x S=%d x D=384 (ViT-S/8 descriptors), randn inputs" % (KNN_Q, KNN_S), "ms_per_sweep": round(p3 * 1e3, 3),
                       "queries_per_s": round(KNN_Q / p3), "algorithmic_bytes": KNN_Q * 384 * 4 + KNN_S * 384 * 4 + KNN_Q * 80,
                       "algorithmic_GBps": round((KNN_Q * 384 * 4 + KNN_S * 384 * 4 + KNN_Q * 80) / p3 / 1e9, 1)}
        del sh3, q3
    if c.rank == 0 and not c.use_dist and sample_check:
        from oracle import knn as oknn

        for name, t in ((only or "randn", top), ("structured", top_s)):
            shots, q = inputs[name]
            ref = oknn.dense(cls.cpu(), shots.cpu(), q[:sample_check].cpu(), True)
            same = "%d / %d" % (int((t[:sample_check].cpu() == ref).all(dim=1).sum()), sample_check)
            if name != "structured" or only == "structured":
                out["top10_rows_identical_to_oracle"] = same
            else:
                out["structured"]["top10_rows_identical_to_oracle"] = same
    return out


# ----------------------------------------------------------------------------------------------- training (cfg 3)
def _train_batch(c, batch_per_gpu, num_classes, with_proposals=False):
    import torch

    from lvc_amd.structures import Boxes, Instances
    from lvc_amd.utils import synthetic as syn

    g = torch.Generator().manual_seed(1 + c.rank)
    batch = []
    for i in range(batch_per_gpu):
        h, w, n = 800, 1333, 8
        x1 = torch.rand(n, generator=g) * (w - 300)
        y1 = torch.rand(n, generator=g) * (h - 300)
        bw = 40 + torch.rand(n, generator=g) * 250
        bh = 40 + torch.rand(n, generator=g) * 250
        boxes = torch.stack([x1, y1, x1 + bw, y1 + bh], 1)
        inst = Instances((h, w))
        inst.gt_boxes = Boxes(boxes)
        inst.gt_classes = torch.randint(0, num_classes, (n,), generator=g)
        d = {"image": syn.synthetic_image(1 + (c.rank * batch_per_gpu + i) % 16).to(c.dev), "instances": inst, "height": h, "width": w}
        if with_proposals:     # the box corrector trains on jittered ground-truth boxes handed over as proposals (LOAD_PROPOSALS)
            props = Instances((h, w))
            props.proposal_boxes = Boxes(boxes.repeat(8, 1) + torch.randn(8 * n, 4, generator=g) * 10)
            props.objectness_logits = torch.zeros(8 * n)
            d["proposals"] = props
        batch.append(d)
    return batch


def train_leg(c, steps, warmup, batch_per_gpu=8, which="cfg3", phases=True):
    """which = "cfg3": BASELINE configs[2], COCO 30-shot novel fine-tune (only the box predictor trains: 4 tensors, 0.41 MB of
    gradients), R50-FPN, 8 images of 800x1333 per GPU, gradients averaged with ONE flattened all-reduce over RCCL
    (lvc/engine/defaults.py:326-331), SGD step (detectron2/engine/train_loop.py:211-250).
    which = "cfg5": BASELINE configs[4], the box corrector (tools/train_net_reg.py: CascadeROIHeads + BoxOnlyLayersCascade on RBG
    proposals, cascade_ubbr_R_101_FPN_base.yaml: FREEZE_AT 2, 133 trainable tensors), R101-FPN, 2 images of 800x1333 per GPU
    (IMS_PER_BATCH 16 over 8 GPUs), gradients through `GradientBuckets` (64 MB buckets launched under the backward).
    `phases`: after the timed steps, three more with a synchronize between forward / backward / optimizer (attribution only)."""
    import torch

    from lvc_amd import distributed as D
    from lvc_amd.config import set_global_cfg
    from lvc_amd.config.presets import base_rcnn_fpn
    from lvc_amd.modeling import build_model
    from lvc_amd.utils import synthetic as syn
    from lvc_amd.utils.events import EventStorage

    if which == "cfg3":
        cfg = base_rcnn_fpn(num_classes=20, device="cuda:%d" % c.local_rank)
        cfg.MODEL.BACKBONE.FREEZE = True
        cfg.MODEL.PROPOSAL_GENERATOR.FREEZE = True
        cfg.MODEL.ROI_HEADS.FREEZE_FEAT = True
        depth, ncls = 50, 20
    else:
        cfg = base_rcnn_fpn(depth=101, num_classes=60, device="cuda:%d" % c.local_rank)
        M = cfg.MODEL
        M.ROI_HEADS.NAME = "CascadeROIHeads"; M.ROI_HEADS.OUTPUT_LAYER = "BoxOnlyLayersCascade"
        M.ROI_HEADS.PROPOSAL_APPEND_GT = False; M.ROI_HEADS.POSITIVE_FRACTION = 1.0
        M.ROI_HEADS.BATCH_SIZE_PER_IMAGE = 64; M.ROI_HEADS.IOU_THRESHOLDS = [0.3]
        M.ROI_BOX_HEAD.NUM_FC = 3; M.ROI_BOX_HEAD.CLS_AGNOSTIC_BBOX_REG = True
        M.ROI_BOX_CASCADE_HEAD.IOUS = (0.3, 0.5, 0.7); M.PROPOSAL_GENERATOR.NAME = "RBG"; M.LOAD_PROPOSALS = True
        depth, ncls = 101, 60
    set_global_cfg(cfg)
    model = build_model(cfg)
    syn.conditioned_r50_fpn_(model, depth=depth)
    model.train()
    torch.manual_seed(20 + c.rank)          # per-rank sampling seed = SEED + rank (lvc/engine/defaults.py:198)
    batch = _train_batch(c, batch_per_gpu, ncls, with_proposals=which != "cfg3")
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.SGD(params, lr=1e-3 if which == "cfg3" else 1e-4, momentum=0.9, weight_decay=1e-4)
    buckets = D.GradientBuckets(params) if which != "cfg3" else None
    nbytes = 0

    exch = None      # a list: (event after backward() returned, event after the exchange finished) per step, on the compute stream

    def step(sync=None):
        nonlocal nbytes
        t = [time.perf_counter()]
        losses = model(batch)
        if sync is not None:
            torch.cuda.synchronize(); t.append(time.perf_counter())
        opt.zero_grad()      # set_to_none: the deferred weight gradients are written straight into the bucket slices
        sum(losses.values()).backward()
        if exch is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        if buckets is not None:
            nbytes = buckets.finish()
        else:
            nbytes = D.allreduce_gradients_(params)
        if exch is not None:
            e1.record()
            exch.append((e0, e1))
        if sync is not None:
            torch.cuda.synchronize(); t.append(time.perf_counter())
        opt.step()
        if sync is not None:
            torch.cuda.synchronize(); t.append(time.perf_counter())
            sync.append([b - a for a, b in zip(t, t[1:])])
        return losses

    with EventStorage(0):
        for _ in range(max(1, warmup)):
            losses = step()
        _barrier(c)
        t0 = time.perf_counter()
        for _ in range(steps):
            losses = step()
        _barrier(c)
        dt, every = _max_and_all(c, time.perf_counter() - t0)
        # gradient exchange: how much of it the backward hides.  exposed = compute-stream time between the end of
        # backward() and the end of the exchange (what the step pays); alone = the same collectives back to back on idle GPUs;
        # hidden = alone - exposed (cfg 5's buckets start under the backward; cfg 3's one flat all-reduce has nothing to hide under)
        exch = []
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        exposed_ms = sum(a.elapsed_time(b) for a, b in exch) / len(exch)
        exch = None
        _barrier(c)
        t1 = time.perf_counter()
        if buckets is not None:
            for b in buckets.buckets:
                b["pending"] = 0
            buckets.finish()
        else:
            D.allreduce_gradients_(params)
        torch.cuda.synchronize()
        alone_ms = 1e3 * (time.perf_counter() - t1)
        exchange = _gather_obj(c, {"exposed_ms": round(exposed_ms, 3), "alone_ms": round(alone_ms, 3),
                                   "hidden_ms": round(max(0.0, alone_ms - exposed_ms), 3)})
        ph = []
        if phases:
            for _ in range(3):
                step(sync=ph)
    # Roofline of the backward's two kernel families: one more step with a HIP-event pair around every weight- and
    # data-gradient launch (kernels.BWD_TIMER; the conv / GEMM launches of the backward, not the elementwise glue)
    bwd_roof = None
    if phases and which != "cfg3" and c.rank == 0:
        from lvc_amd import kernels as K

        K.BWD_TIMER = []
        try:
            with EventStorage(0):
                step()
            torch.cuda.synchronize()
            rec, K.BWD_TIMER = K.BWD_TIMER, None
        finally:
            K.BWD_TIMER = None
        bwd_roof = {}
        for kind in ("wgrad", "dgrad"):
            rs = [r for r in rec if r[0] == kind]
            if not rs:
                continue
            ms = sum(r[4].elapsed_time(r[5]) for r in rs)
            fl, nb = sum(r[2] for r in rs), sum(r[3] for r in rs)
            eng = sorted({r[1] for r in rs})
            peak = PEAK_F16X2_TFLOPS if eng == ["f16x2"] else PEAK_BF16X3_TFLOPS if "f32" not in eng else PEAK_F32_MFMA_TFLOPS
            bwd_roof[kind] = {"launches": len(rs), "ms": round(ms, 3), "algorithmic_gflop": round(fl / 1e9, 1), "tflops": round(fl / ms / 1e9, 1),
                              "engine": "+".join(eng), "peak_tflops": round(peak, 1), "frac_of_mfma_peak": round(fl / ms / 1e9 / peak, 4),
                              "algorithmic_bytes": int(nb), "algorithmic_GBps": round(nb / ms / 1e6, 1),
                              "frac_of_hbm_peak": round(nb / ms / 1e6 / PEAK_HBM_GBPS, 4)}
        bwd_roof["note"] = ("HIP events around every lvc_conv_wgrad_* / data-gradient conv launch of ONE extra step (weight gradients: bf16 three-way "
                            "split, 6 MFMAs per product -> peak 2500 / 6; data gradients run on the forward kernels with flipped weights); algorithmic "
                            "bytes = operands + result once")
    chk = torch.cat([p.detach().reshape(-1) for p in params]).double().sum()
    same = None
    if c.use_dist:      # after averaged gradients + identical SGD steps every rank must hold the same parameters
        import torch.distributed as dist

        lo, hi = chk.clone(), chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        same = bool(float(hi - lo) == 0.0)
    name = ("cfg 3 novel fine-tune step (fwd + bwd of the box predictor + RCCL gradient all-reduce + SGD), R50-FPN" if which == "cfg3" else
            "cfg 5 box-corrector step (tools/train_net_reg.py: fwd + bwd through res3..res5 / FPN / cascade heads + bucketed RCCL gradient exchange + SGD), R101-FPN")
    out = {"workload": "%s, %d x 800x1333 per GPU" % (name, batch_per_gpu), "trainable_tensors": len(params),
           "value": round(c.world * batch_per_gpu * steps / dt, 2), "unit": "img/s", "ms_per_step": round(1e3 * dt / steps, 3),
           "global_batch": batch_per_gpu * c.world, "gradient_bytes_allreduced": nbytes,
           "gradient_exchange": {"per_rank": exchange, "mode": "one flattened all-reduce after backward()" if buckets is None else
                                 "64 MB buckets, all-reduce launched as each bucket's last gradient arrives (GradientBuckets)",
                                 "note": "exposed = compute-stream ms from the end of backward() to the end of the exchange; alone = the same "
                                         "collectives with nothing else running; hidden = alone - exposed.  World size 1: no collective runs"},
           "per_rank_img_per_s": [round(batch_per_gpu * steps / t, 2) for t in every],
           "parameters_identical_across_ranks": same, "losses": {k: round(float(v.detach()), 5) for k, v in losses.items()}}
    if ph:
        out["ms_forward_backward_optimizer"] = [round(1e3 * sum(p[i] for p in ph) / len(ph), 2) for i in range(3)]
        out["phase_note"] = "3 extra steps with a synchronize after each phase (attribution; the timed steps have none)"
    if bwd_roof:
        out["roofline"] = bwd_roof
    if buckets is not None:
        buckets.remove()
    del model, opt
    torch.cuda.empty_cache()
    return out


def vendor_trunk_leg(timeout_s=240):
    """Context, not a product path and not `vs_baseline`: the same R50-FPN trunk (stem, res2-5, FPN) on PyTorch-ROCm's own convolutions
    (F.conv2d = MIOpen, what the reference's modules would call on this GPU) in fp32, against this repo's backbone on the same batch
    (scripts/probe_torch_trunk.py in a subprocess: the library's first call searches its kernels for ~20 s)."""
    import re

    env = dict(os.environ, LVC_TRUNK_YARDSTICK_FP32_ONLY="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "probe_torch_trunk.py")], cwd=ROOT, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.DEVNULL, timeout=timeout_s, text=True)
    mine = re.search(r"lvc_amd backbone \+ FPN.*?: ([0-9.]+) ms", r.stdout)
    lib = re.search(r"F\.conv2d fp32, channels_last: ([0-9.]+) ms per batch  \(([0-9.]+) x lvc_amd; max \|diff\| / scale vs lvc_amd ([0-9.e+-]+)\)", r.stdout)
    if not (mine and lib):
        return {"error": "no result (rc %d)" % r.returncode}
    return {"workload": "R50-FPN trunk (stem, res2-5, FPN + p6), 8 x 800x1333, FrozenBN folded", "lvc_amd_ms": float(mine.group(1)),
            "pytorch_rocm_miopen_fp32_ms": float(lib.group(1)), "ratio": float(lib.group(2)), "max_abs_diff_over_scale": float(lib.group(3)),
            "note": "yardstick only: F.conv2d (MIOpen as shipped in the image, channels_last, default find mode) on the same GPU in the same run; "
                    "fp16 through the same library: profiles/r05_torch_trunk.txt"}


def descriptor_leg(c, batch=64, steps=5, warmup=2):
    """the descriptor network in front of the kNN sweep (tools/run_nearest_neighbours.py:102-128 get_descriptors,
    :292-293 DINO ViT-S/8): crops/s on 224x224 crops at batch 64, random-init weights of that architecture.  Algorithmic work per
    crop: 12 blocks x (qkv + proj + fc1 + fc2 over 785 tokens x 384 channels + two 785 x 785 x 64 products per head) + the patch
    embedding."""
    import torch

    from lvc_amd import kernels as K
    from lvc_amd.modeling.vit import seeded_state_dict_, vit_small

    m = vit_small(8)
    seeded_state_dict_(m, 0)
    m = m.to(c.dev).eval()
    g = torch.Generator(device=c.dev).manual_seed(3)
    x = torch.rand(batch, 3, 224, 224, device=c.dev, generator=g) * 255.0
    mean, std = [123.675, 116.28, 103.53], [58.395, 57.12, 57.375]
    N, D, Hd = 785, 384, 6
    gflop = (12 * (2 * N * D * (3 * D + D + 4 * D + 4 * D) + 4 * Hd * N * N * 64) + 2 * 784 * 192 * D) / 1e9
    with torch.no_grad():
        for _ in range(warmup):
            y = m(x, mean, std)
        _barrier(c)
        t0 = time.perf_counter()
        for _ in range(steps):
            y = m(x, mean, std)
        _barrier(c)
        dt = (time.perf_counter() - t0) / steps
        qkv = torch.randn(batch * N, 3 * D, device=c.dev, generator=g)
        att = _event_ms(lambda: K.mha(qkv, batch, N, Hd, 64, 0.125), 5)
    K.check_conv_error_word(c.dev)
    out = {"workload": "ViT-S/8 descriptors, %d crops of 3x224x224 per batch (785 tokens, 12 blocks, 6 heads)" % batch,
           "value": round(batch / dt, 1), "unit": "crops/s", "ms_per_batch": round(dt * 1e3, 3), "algorithmic_gflop_per_crop": round(gflop, 2),
           "tflops": round(gflop * batch / dt / 1e3, 1), "frac_of_f16x2_peak": round(gflop * batch / dt / 1e3 / PEAK_F16X2_TFLOPS, 4),
           "attention_kernel": {"kernel": "mha_mfma_kernel (+ mha_split_kernel): softmax(q k^T / 8) v for %d x %d heads of 785 x 64" % (batch, Hd),
                                "ms_per_layer": round(att, 4), "tflops": round(4.0 * batch * Hd * N * N * 64 / att / 1e9, 1),
                                "frac_of_f16x2_peak": round(4.0 * batch * Hd * N * N * 64 / att / 1e9 / PEAK_F16X2_TFLOPS, 4)},
           "descriptor_norm_finite": bool(torch.isfinite(y).all())}
    del m, x, qkv
    torch.cuda.empty_cache()
    return out


def r101_leg(c, steps=10, warmup=3, parity_images=2):
    """R101-FPN inference at the headline's batch (the depth BASELINE configs[4] names), same timing rules."""
    import torch

    from lvc_amd.config.presets import base_rcnn_fpn
    from lvc_amd.modeling import build_model
    from lvc_amd.utils import synthetic as syn

    model = build_model(base_rcnn_fpn(depth=101, device="cuda:%d" % c.local_rank)).eval()
    syn.conditioned_r50_fpn_(model, depth=101)
    batch = [{"image": syn.synthetic_image(1 + (c.rank * BATCH_PER_GPU + i) % 16).to(c.dev), "height": 800, "width": 1333} for i in range(BATCH_PER_GPU)]
    with torch.no_grad():
        for _ in range(warmup):
            out = model.inference_batched(batch)
        _barrier(c)
        t0 = time.perf_counter()
        for _ in range(steps):
            out = model.inference_batched(batch)
        _barrier(c)
        dt, _ = _max_and_all(c, time.perf_counter() - t0)
    n_det = out[3].tolist()
    parity = None
    if c.rank == 0 and parity_images > 0:
        # the LAST TIMED step's detections of the first images against the CPU oracle at depth 101 (pinned against the reference's own
        # R101 run at this size: tests/test_oracle_golden.py::test_e2e_r101_800x1333), through the same noise gate as the headline
        from oracle import noise as onoise
        from oracle import rcnn as orc

        torch.set_num_threads(min(os.cpu_count() or 1, 32))
        ob, osc, ocl, cnt = (t.cpu() for t in out[:4])
        idx = list(range(min(parity_images, BATCH_PER_GPU)))
        hip = [(ob[i, : int(cnt[i])], osc[i, : int(cnt[i])], ocl[i, : int(cnt[i])].long()) for i in idx]
        sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        spec = orc.RCNNSpec(depth=101)
        cpu_in = [{"image": batch[i]["image"].cpu(), "height": 800, "width": 1333} for i in idx]
        t1 = time.perf_counter()
        with torch.no_grad():
            ref = orc.generalized_rcnn_inference(sd, spec, cpu_in)
            nz = onoise.fp32_vs_fp64(sd, spec, cpu_in, res32=ref, derive_identity=True)
        dev = onoise.deviation(hip, [(r["pred_boxes"], r["scores"], r["pred_classes"]) for r in ref], nz["box_tol"], nz["score_tol"])
        ok, bars, msg = onoise.gate(dev, nz)
        parity = {"images_checked": idx, "deviation_among_matched": {k: (round(v, 6) if isinstance(v, float) else v) for k, v in dev.items()},
                  "cpu_path_fp32_vs_fp64_noise": {k: (round(v, 6) if isinstance(v, float) else v) for k, v in nz.items()},
                  "bars": {k: round(v, 6) for k, v in bars.items()}, "gate": msg, "gate_ok": bool(ok), "seconds": round(time.perf_counter() - t1, 1),
                  "pass_bar": "as timed_batch_parity, with the identity bars themselves derived from the noise run (%g x its medians of a first, generous "
                              "matching: R101's conditioned weights carry ~6x R50's fp32 noise): equal counts, found fraction >= the CPU path's own - %g, "
                              "median / p90 <= %g x and the largest <= %g x its fp32-vs-fp64 noise on these images; else the run exits non-zero"
                              % (onoise.IDENT_K, onoise.IDENT_MARGIN, onoise.K_NOISE, onoise.K_MAX)}
    del model
    torch.cuda.empty_cache()
    return {"workload": "R101-FPN GeneralizedRCNN inference, bs=%d synthetic 3x800x1333 per GPU" % BATCH_PER_GPU,
            "value": round(c.world * BATCH_PER_GPU * steps / dt, 2), "unit": "img/s", "ms_per_step": round(1e3 * dt / steps, 3),
            "detections_per_image": n_det, "parity": parity}


def eval_loop_leg(c, model, steps, warm=4, distinct=3):
    """The evaluation loop as the reference times it (lvc/evaluation/evaluator.py:85-157): batches arrive as HOST tensors from a
    loader, `lvc_amd.evaluation.inference_on_dataset` (two batches in flight) moves them to the GPU on the stream the batch runs
    on -- the copy of batch i+1 overlaps the trunk of batch i -- and every batch's Instances are collected (one D2H read each).
    Three input forms: {"image": uint8 CHW 3x800x1333} -- what the reference's DatasetMapper hands over (lvc/data/dataset_mapper.py:164:
    `torch.as_tensor(image.transpose(2, 0, 1))` of the uint8 array its ResizeShortestEdge produced; 3.2 MB per image across PCIe) --,
    {"image": float32 CHW 3x800x1333} (a mapper that converts on the host: 12.8 MB per image) and {"raw": uint8 HWC 480x800x3} (the
    decoded file: ResizeShortestEdge on the device, Pillow-exact, to 800x1333; 1.15 MB per image).  Host tensors are pinned, as a
    DataLoader(pin_memory=True) delivers them."""
    import torch

    from lvc_amd.evaluation import inference_on_dataset
    from lvc_amd.utils import synthetic as syn

    out = {}
    for kind in ("image_u8", "image", "raw"):
        batches = []
        for b in range(distinct):
            items = []
            for i in range(BATCH_PER_GPU):
                seed = 1 + (c.rank * BATCH_PER_GPU + b * BATCH_PER_GPU + i) % 16
                if kind == "image_u8":
                    items.append({"image": syn.synthetic_image(seed).round().clamp(0, 255).to(torch.uint8).pin_memory(), "height": 800, "width": 1333})
                elif kind == "image":
                    items.append({"image": syn.synthetic_image(seed).pin_memory(), "height": 800, "width": 1333})
                else:
                    raw = syn.synthetic_image(seed, 480, 800).permute(1, 2, 0).round().clamp(0, 255).to(torch.uint8).contiguous()
                    items.append({"raw": raw.pin_memory(), "height": 480, "width": 800})
            batches.append(items)
        key = "raw" if kind == "raw" else "image"
        nbytes = sum(x[key].numel() * x[key].element_size() for x in batches[0])

        def loader(n):
            for i in range(n):
                yield batches[i % distinct]

        for _ in inference_on_dataset(model, loader(warm), depth=2):
            pass
        _barrier(c)
        t0 = time.perf_counter()
        n_img = n_det = 0
        for _inputs, outputs in inference_on_dataset(model, loader(steps), depth=2):
            n_img += len(outputs)
            n_det += sum(len(o["instances"]) for o in outputs)
        _barrier(c)
        dt, _ = _max_and_all(c, time.perf_counter() - t0)
        out[kind] = {"value": round(c.world * n_img / dt, 2), "unit": "img/s", "ms_per_step": round(1e3 * dt / steps, 3),
                     "host_bytes_per_batch": nbytes, "pcie_GBps": round(nbytes * steps / dt / 1e9, 2), "detections_collected": n_det,
                     "input": "uint8 CHW 3x800x1333 (the reference DatasetMapper's output, dataset_mapper.py:164), pinned host memory" if kind == "image_u8"
                     else "float32 CHW 3x800x1333, pinned host memory" if kind == "image"
                     else "uint8 HWC 480x800x3 (decoded file), pinned host memory; ResizeShortestEdge(800, 1333) on the device"}
    out["note"] = ("inference_on_dataset(depth=2) over %d batches of %d HOST-resident images, every batch's Instances collected; H2D copies run on "
                   "the batch's own stream and overlap the other stream's trunk.  `value` (the headline) starts with its inputs resident in HBM."
                   % (steps, BATCH_PER_GPU))
    return out


# ----------------------------------------------------------------------------------------------- HBM-side kernels
def bandwidth_kernels(c, model, batch):
    """ROIAlign and batched NMS on the tensors of a real step, each launched alone between HIP events."""
    import torch

    from lvc_amd import kernels as K

    out = {}
    with torch.no_grad():
        images = model.preprocess_image(batch)
        sizes_dev = model._dev_const(images.image_sizes, torch.int32)
        N, _, Hp, Wp = images.tensor.shape
        x4 = images.tensor.as_strided((N, Hp, Wp, 4), (Hp * Wp * 4, Wp * 4, 4, 1), images.tensor.storage_offset())
        feats = model.backbone.forward_nhwc(x4)
        pg = model.proposal_generator
        pboxes, _pl, pcount = pg.predict_proposals_batched(feats, sizes_dev)
        flist = [feats[f] for f in model.roi_heads.in_features]
        pooler = model.roi_heads.box_pooler
        ms = _event_ms(lambda: pooler.pool_nhwc(flist, pboxes), 20)
        R = pboxes.shape[1]
        C = flist[0].shape[3]
        wr = N * R * 49 * C * 4
        rd = sum(f.shape[1] * f.shape[2] for f in flist) * C * 4 * N
        alg = wr + rd + N * R * 20
        out["roi_align"] = {"kernel": "roi_align_fwd (all levels, one launch; lvc_assign_levels_rois included)", "ms": round(ms, 4),
                            "algorithmic_bytes": alg, "bytes_note": "%d RoIs x 7x7x%d fp32 written + each pyramid byte p2..p5 read once + rois (SURVEY 8(d): <=141.6 MB/img)" % (N * R, C),
                            "GBps": round(alg / ms / 1e6, 1), "frac_of_hbm_peak": round(alg / ms / 1e6 / PEAK_HBM_GBPS, 4)}
        # batched NMS alone, on the step's own pre-NMS candidates: per level the top-1000 decoded boxes in score order
        # (lvc_rpn_proposals with the suppression switched off), concatenated with their level ids
        fused = pg.rpn_head.forward_nhwc([feats[f] for f in pg.in_features])
        A = pg.rpn_head.num_anchors
        cell = list(pg.anchor_generator.cell_anchors)
        bs, ss, ls = [], [], []
        for l, f in enumerate(fused):
            n = min(1000, f.shape[1] * f.shape[2] * A)
            b, s, _ = K.rpn_proposals([f[..., :A]], [f[..., A:5 * A]], [cell[l]], [pg.anchor_generator.strides[l]], sizes_dev,
                                      1000, n, 1.0, 0.0)
            bs.append(b)
            ss.append(s)
            ls.append(torch.full((N, n), l, dtype=torch.int32, device=c.dev))
        cb, cs, cl = torch.cat(bs, 1).contiguous(), torch.cat(ss, 1).contiguous(), torch.cat(ls, 1).contiguous()
        ncand = cb.shape[1]
        ms = _event_ms(lambda: K.batched_nms_batch(cb, cs, cl, None, 0.7, 1000), 20)
        keep, nk = K.batched_nms_batch(cb, cs, cl, None, 0.7, 1000)
        alg = N * ncand * (16 + 4 + 8) + N * ncand * 8
        out["batched_nms"] = {"kernel": "lvc_batched_nms (sort prep + ballot mask + reduce), %d images x %d candidates with level ids, thr 0.7" % (N, ncand),
                              "ms": round(ms, 4), "kept_per_image": nk.tolist(), "algorithmic_bytes": alg,
                              "bytes_note": "28 N in + 8 N out per image (SURVEY 8(d)); the N x N/64 x 8 B mask is implementation traffic",
                              "GBps": round(alg / ms / 1e6, 2), "frac_of_hbm_peak": round(alg / ms / 1e6 / PEAK_HBM_GBPS, 6),
                              "pair_tests_per_s": round(N * ncand * ncand / 2 / (ms * 1e-3))}
        ms = _event_ms(lambda: pg.predict_proposals_batched(feats, sizes_dev, fused=fused), 20)
        out["rpn_proposals"] = {"kernel": "lvc_rpn_proposals: per-level radix top-k + decode + per-(image, level) NMS + rank merge (the form the detector runs)",
                                "ms": round(ms, 4)}
    return out


# ----------------------------------------------------------------------------------------------- inference (cfg 2)
def infer_main(c, args):
    import torch
    import torch.distributed as dist

    from lvc_amd import kernels as K
    from lvc_amd.config.presets import base_rcnn_fpn
    from lvc_amd.modeling import build_model
    from lvc_amd.utils import synthetic as syn

    cfg = base_rcnn_fpn(depth=50, num_classes=80, device="cuda:%d" % c.local_rank)
    model = build_model(cfg).eval()
    syn.conditioned_r50_fpn_(model)
    # the rank's shard of the (synthetic) dataset: contiguous block, like reference InferenceSampler
    imgs = [syn.synthetic_image(1 + (c.rank * BATCH_PER_GPU + i) % 16).to(c.dev) for i in range(BATCH_PER_GPU)]
    batch = [{"image": im, "height": 800, "width": 1333} for im in imgs]

    def step():
        with torch.no_grad():
            return model.inference_batched(batch)

    def step_forward():
        # the drop-in surface (reference lvc/modeling/meta_arch/rcnn.py:100-125): list[{"image", ...}] -> list[{"instances": Instances}];
        # = inference_batched + the step's ONE device->host read (counts, status, range words) + per-image slicing.  This is `value`.
        with torch.no_grad():
            return model(batch)

    for _ in range(args.warmup):
        res = step_forward()
    _barrier(c)
    # HIP events on the launch stream: the timed region brackets only the launches of the DOMINANT kernel (picked from
    # one fully bracketed untimed step; ~150 event records per step cost ~0.7 ms of host time inside the timed region),
    # the per-kernel breakdown of every conv/GEMM launch comes from two more untimed steps after it.
    timer = full = None
    NAMES = {"f16x2_halo": "conv3x3_halo_s1_kernel (pipelined 3x3: one accumulator in the trunk, two in the RPN head)" if K.HALO_S1 else "conv3x3_halo_h2_kernel", "f16x2_pw": "conv_pw_dma_kernel (LDS-DMA pointwise: the layers with fewer than 64 input or output channels)",
             "f16x2_wino": "conv3x3_wino_kernel (Winograd F(2,3) along x: the 3x3 layers on the large maps, 6 products per output instead of 9)",
             "f16x2_pws1": "conv_pw_s1_kernel (pipelined pointwise / FC: every layer with >= 64 input and output channels)", "bf16x3_halo": "conv3x3_halo_kernel",
             "f16s1_chain": "conv_pw_chain_kernel (conv3 + shortcut add + ReLU -> the next block's conv1 in one launch: res2 / res3)",
             "f16s1_bneck": "conv_bneck_kernel (a whole res2 bottleneck block -- conv1, 3x3, conv3 + shortcut + ReLU -- in one launch)",
             "bf16x3": "conv_bf16x3_kernel (+ bf16 pointwise shapes)", "f32": "conv_igemm_f32_kernel"}
    if not args.no_launch_timer:
        probe = K.LaunchTimer()
        K.CONV_TIMER = probe
        step()
        K.CONV_TIMER = None
        dom = max(NAMES, key=lambda e: probe.flops_and_ms(e)[1])
        # ... and only in every 4th step of the timed region: an event pair opens ~6 us of stream bubbles around the launch it
        # brackets (22 launches per step: 0.27 ms = 1.8 % of a step when every step is instrumented, rocprofv3 kernel trace)
        timer = K.LaunchTimer(only={dom}, every=4)
        _barrier(c)
    K.CONV_TIMER = timer
    cpu0 = time.thread_time()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        if timer is not None:
            timer.next_step()
        res = step_forward()
    host_cpu = time.thread_time() - cpu0      # this rank's launching thread, CPU time (not the wait for the step's one device->host read)
    _barrier(c)
    dt = time.perf_counter() - t0
    K.CONV_TIMER = None
    dt_max, dt_all = _max_and_all(c, dt)
    # what the host must supply per step and rank: the time to ENQUEUE a step (inference_batched returns without a device->host read)
    enq = []
    for _ in range(3):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        step()
        enq.append(time.perf_counter() - t1)
    torch.cuda.synchronize()
    host_us_all = [round(h, 1) for h in _gather_obj(c, 1e6 * min(enq))]
    host_cpu_all = [round(h, 1) for h in _gather_obj(c, 1e6 * host_cpu / args.steps)]
    if timer is not None:
        full = K.LaunchTimer()
        K.CONV_TIMER = full
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        K.CONV_TIMER = None
    from lvc_amd.modeling.roi_heads.roi_heads import check_status

    out = step()      # the raw device outputs of the same batch (what forward() slices): parity check and detection counts below
    check_status(int(out[4].item()))
    K.check_conv_error_word(c.dev)   # stream-K timeout / fp16x2 operand-range word of the conv kernels
    n_det = out[3].tolist()
    assert [len(r["instances"]) for r in res] == n_det

    # the same K steps WITHOUT the host read: inference_batched returns device tensors, nothing synchronises inside the loop
    with torch.no_grad():
        _barrier(c)
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        _barrier(c)
        fdt, _ = _max_and_all(c, time.perf_counter() - t1)
    through_forward = {"value": round(c.world * BATCH_PER_GPU * args.steps / fdt, 2), "unit": "img/s",
                       "ms_per_step": round(1e3 * fdt / args.steps, 3),
                       "note": "same K steps through GeneralizedRCNN.inference_batched (device tensors out, no device->host read inside the loop): what "
                               "`value` was in rounds 1-3; `value` is now the reference-signature forward(batched_inputs) -> list[{'instances': Instances}]"}

    # Extra, separately timed pass (not `value`): the same K steps issued round-robin on two HIP streams
    # (lvc_amd/evaluation.py): the latency-bound tail of batch i overlaps the trunk of batch i+1.  Kept out of the timed
    # region above because overlapping launches stretch every kernel's start-to-end time, which is what `roofline` reports.
    pipelined = None
    if args.pipeline_depth > 1:
        from lvc_amd.evaluation import PipelinedInference

        pipe = PipelinedInference(model, args.pipeline_depth)
        for _ in range(2 * args.pipeline_depth):
            pipe.submit(batch, collectable=False)
        _barrier(c)
        t1 = time.perf_counter()
        for _ in range(args.steps):
            pipe.submit(batch, collectable=False)
        _barrier(c)
        pdt = time.perf_counter() - t1
        pipe.synchronize()
        pmax, _ = _max_and_all(c, pdt)
        pipelined = {"streams": args.pipeline_depth, "value": round(c.world * BATCH_PER_GPU * args.steps / pmax, 2),
                     "unit": "img/s", "ms_per_step": round(1e3 * pmax / args.steps, 3),
                     "note": "same K steps, batches in flight on 2 HIP streams -- what lvc_amd.evaluation.inference_on_dataset does by default (depth=2); results bit-identical (tests/test_gpu_pipeline.py); not `value` because overlapping launches stretch the per-launch times `roofline` is computed from"}

    # The same K steps as ONE hipGraph launch each (lvc_amd.evaluation.GraphedInference): what the step costs when the host's
    # launch loop is taken out of it (eight ranks share one host at N = 8).  Not `value`.
    graphed = None
    if not args.no_extras:
        try:
            from lvc_amd.evaluation import GraphedInference

            gi = GraphedInference(model, batch)
            for _ in range(3):
                gi.replay()
            _barrier(c)
            t1 = time.perf_counter()
            for _ in range(args.steps):
                gi.replay()
            _barrier(c)
            gdt, _ = _max_and_all(c, time.perf_counter() - t1)
            same = all(bool(torch.equal(a, b)) for a, b in zip(gi.out, out))
            graphed = {"value": round(c.world * BATCH_PER_GPU * args.steps / gdt, 2), "unit": "img/s", "ms_per_step": round(1e3 * gdt / args.steps, 3),
                       "outputs_bit_identical_to_eager": same,
                       "note": "inference_batched captured in one hipGraph and replayed (host cost per step: one graph launch)"}
            del gi
        except Exception as e:
            graphed = {"error": repr(e)}

    # The evaluation loop fed from host memory: not `value` -- the contract's value starts with inputs in HBM
    eval_loop = None
    if not args.no_extras and args.pipeline_depth > 1:
        try:
            eval_loop = eval_loop_leg(c, model, args.steps)
            if pipelined is not None:
                eval_loop["raw_over_pipelined"] = round(eval_loop["raw"]["value"] / pipelined["value"], 4)
                eval_loop["image_over_pipelined"] = round(eval_loop["image"]["value"] / pipelined["value"], 4)
                eval_loop["image_u8_over_pipelined"] = round(eval_loop["image_u8"]["value"] / pipelined["value"], 4)
        except Exception as e:
            eval_loop = {"error": repr(e)}

    total_imgs = c.world * BATCH_PER_GPU * args.steps
    value = total_imgs / dt_max

    roofline = None
    if timer is not None:
        fl, ms, nlaunch = timer.flops_and_ms(dom)
        achieved = fl / (ms * 1e-3) / 1e12
        peak = PEAK_F32_MFMA_TFLOPS if dom == "f32" else PEAK_F16X2_TFLOPS if dom.startswith("f16") else PEAK_BF16X3_TFLOPS
        traffic = pmc = None
        live = step_pmc = None
        if c.rank == 0 and c.world == 1 and not args.no_live_pmc:
            live = _live_conv_pmc()
            step_pmc = _live_step_pmc()
        if live is not None:
            traffic = live["hbm_bytes_per_launch"]
            pmc = live
        else:
            pmc_file = os.path.join(ROOT, "profiles", "r02_conv_pmc.json")
            if not os.path.exists(pmc_file):
                pmc_file = os.path.join(ROOT, "profiles", "r01_conv_pmc.json")
            if os.path.exists(pmc_file):
                pj = json.load(open(pmc_file))
                traffic = pj.get("hbm_bytes_per_launch")
                sq = pj.get("sq_counters_same_launch", {})
                pmc = {"source": os.path.relpath(pmc_file, ROOT) + " (a separate rocprofv3 --pmc pass of one p2 3x3 launch; no live pass in this run)",
                       "mfma_busy_fraction": sq.get("mfma_busy_fraction"), "clock_GHz_under_load": sq.get("clock_GHz")}
        roofline = {
            "kernel": "%s (%d launches/step; HIP-event brackets in %d of the %d timed steps)" % (NAMES[dom], nlaunch // max(1, timer.steps_timed()), timer.steps_timed(), args.steps),
            "bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
            "peak_note": ("2500 TFLOP/s dense fp16 MFMA / 3 MFMAs per fp32-accurate product (two-way fp16 operand split, a1 b1 + a1 b2 + a2 b1 in fp32 accumulators)" if dom.startswith("f16") else "2500 TFLOP/s dense bf16 MFMA / 6 MFMAs per fp32-accurate product (exact 3-way bf16 operand split, fp32 accumulate)") + "; achieved counts algorithmic fp32 flops once (2 M K C 9 for a 3x3 layer -- also for the Winograd F(2,3) kernel, which issues two thirds of the direct form's MFMAs for them: its ceiling in this unit is 1.5 x 833); under the socket power cap a bare loop of these MFMAs on random data sustains 1.66 PFLOP/s = 552 TFLOP/s fp32-equivalent (profiles/r03_mfma_ceiling.txt)",
            "frac_convention": "frac = algorithmic fp32 flops (direct form, counted once) / (2500 / 3): the fp32-accurate ceiling of the two-way fp16 split; the two "
                               "readings against the guide's raw 2500 TFLOP/s follow",
            "frac_issued_mfma_of_2500": round(achieved * (3.0 if dom.startswith("f16") else 6.0) * (2.0 / 3.0 if dom == "f16x2_wino" else 1.0) / 2500.0, 4),
            "frac_algorithmic_of_2500": round(achieved / 2500.0, 4),
            "frac_of_fp32_mfma_peak_157.3": round(achieved / PEAK_F32_MFMA_TFLOPS, 4),
            "traffic": traffic,
            "traffic_source": ("live rocprofv3 --pmc passes in this run" if live is not None else "profiles (not measured in this run)")
                              + ": fabric bytes FETCH_SIZE x2 + WRITE_SIZE of ONE p2 3x3 launch (the dominant kernel's largest: the FPN output conv on p2; the RPN head on p2 is the same shape) vs 1.10 GB algorithmic",
            "mfma_utilisation_pmc": pmc,
            "kernel_ms_per_step": round(ms / max(1, timer.steps_timed()), 3), "launch_avg_ms": round(ms / nlaunch, 4)}
        other = {}
        for e in NAMES:
            f2, m2, n2 = full.flops_and_ms(e)
            if n2:
                pk = PEAK_F32_MFMA_TFLOPS if e == "f32" else PEAK_F16X2_TFLOPS if e.startswith("f16") else PEAK_BF16X3_TFLOPS
                tf = f2 / (m2 * 1e-3) / 1e12
                other[NAMES[e]] = {"launches_per_step": n2 // 2, "ms_per_step": round(m2 / 2, 3), "tflops": round(tf, 2),
                                   "frac_of_engine_peak": round(tf / pk, 4)}
        fl_all, ms_all, n_all = full.flops_and_ms()
        other["all_conv_gemm"] = {"launches_per_step": n_all // 2, "algorithmic_gflop_per_image": round(fl_all / (BATCH_PER_GPU * 2) / 1e9, 1),
                                  "ms_per_step": round(ms_all / 2, 3), "tflops": round(fl_all / (ms_all * 1e-3) / 1e12, 2)}
        other["whole_step"] = {"tflops": round(fl_all / 2 / (dt_max / args.steps) / 1e12, 2),
                               "frac_of_f16x2_peak": round(fl_all / 2 / (dt_max / args.steps) / 1e12 / PEAK_F16X2_TFLOPS, 4)}
        roofline["breakdown_untimed_pass"] = other
        # north_star: MFMA utilisation of the BACKBONE (not of one launch) and the whole step's traffic against its algorithmic bytes
        alg_conv = full.algorithmic_bytes() / 2
        alg_other = BATCH_PER_GPU * (3 * 800 * 1333 * 4 + 800 * 1344 * 16) + BATCH_PER_GPU * (1000 * 49 * 256 * 4 + 91.4e6)   # preprocess in / out; ROIAlign out + pyramid once
        roofline["step_algorithmic_bytes"] = int(alg_conv + alg_other)
        roofline["backbone_mfma_busy_target"] = 0.70      # north_star's figure; `backbone_mfma_busy` below is the measured one
        if step_pmc is not None:
            roofline["backbone_mfma_busy"] = step_pmc["backbone_mfma_busy"]
            roofline["step_traffic"] = step_pmc["step_hbm_bytes"]
            roofline["step_traffic_over_algorithmic"] = round(step_pmc["step_hbm_bytes"] / (alg_conv + alg_other), 3)
            roofline["step_pmc"] = step_pmc
        else:
            roofline["backbone_mfma_busy"] = roofline["step_traffic"] = None
            roofline["step_pmc"] = "not measured in this run (--no-live-pmc or rocprofv3 unavailable); the committed pass: profiles/r04_step_pmc.json"

    extras = {}
    if c.rank == 0 and not args.no_extras:
        try:
            extras["bandwidth_kernels"] = bandwidth_kernels(c, model, batch)
        except Exception as e:   # an extra must never cost the headline line
            extras["bandwidth_kernels"] = {"error": repr(e)}
    if c.rank == 0 and c.world == 1 and not args.no_extras:
        try:
            extras["range_reroutes"] = range_rehearsal(c, max(5, args.steps // 2), 3)
        except Exception as e:
            extras["range_reroutes"] = {"error": repr(e)}
    if not args.no_extras:
        try:
            if c.world == 1:
                if c.rank == 0:
                    extras["knn"] = knn_leg(c, 5, 2)
                    bk = extras.get("bandwidth_kernels")
                    if isinstance(bk, dict) and "error" not in bk:
                        bk.update(_knn_kernels_alone(c))
            else:
                extras["dp_legs"] = {"knn": knn_leg(c, 3, 1)}
                extras["dp_legs"]["train_cfg3"] = train_leg(c, 3, 1, phases=False)
        except Exception as e:
            extras["dp_legs_error"] = repr(e)
        # BASELINE configs[2] / [4] are training workloads and configs[4] names R101: their one-GPU rates, every run
        if c.world == 1:
            for key, fn in (("train_cfg3", lambda: train_leg(c, 5, 2, 8, "cfg3")), ("train_cfg5_r101", lambda: train_leg(c, 5, 2, 2, "cfg5")),
                            ("r101_inference", lambda: r101_leg(c, parity_images=0 if args.no_cpu_baseline else 2)), ("descriptors", lambda: descriptor_leg(c)),
                            ("vendor_trunk_yardstick", vendor_trunk_leg)):
                try:
                    extras.setdefault("train", {})[key] = fn()
                except Exception as e:
                    extras.setdefault("train", {})[key] = {"error": repr(e)}

    cpu_baseline = parity = None
    if c.rank == 0 and args.gpus == 1 and not args.no_cpu_baseline:
        ob, osc, ocl, cnt = out[0].cpu(), out[1].cpu(), out[2].cpu(), n_det
        gpu_dets = [(ob[i, :cnt[i]], osc[i, :cnt[i]], ocl[i, :cnt[i]].long()) for i in range(BATCH_PER_GPU)]
        cpu_baseline, parity = _cpu_baseline(model, imgs, gpu_dets, _gpu_proposals(model, batch))

    if c.rank == 0:
        line = {
            "metric": "img/s COCO 800x1333 R50-FPN inference (GeneralizedRCNN.forward(batched_inputs) -> Instances, 1000 proposals, 100 detections)",
            "value": round(value, 2), "unit": "img/s", "n_gpus": c.world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt_max / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32 (conv/GEMM inner products on the fp16 / bf16 matrix cores through fp32-accurate operand splits: two-way fp16 with main + cross fp32 accumulators for the 3x3 and wide 1x1 / FC layers, three-way bf16 for the rest; final boxes/scores closer to the fp64 evaluation than the reference's fp32 CPU path, tests/test_gpu_chain.py)", "data": "synthetic",
            "config": {"workload": "COCO-detection R50-FPN inference, bs=8 synthetic 3x800x1333 per GPU, 1000 pre/post-NMS "
                                   "proposals per level/image, 80 classes, conditioned random-init weights",
                       "batch_per_gpu": BATCH_PER_GPU, "global_batch": BATCH_PER_GPU * c.world, "parallelism": "dp%d" % c.world,
                       "detections_per_image": n_det},
            "rccl": c.rccl, "per_rank": {"img_per_s": [round(BATCH_PER_GPU * args.steps / t, 2) for t in dt_all],
                                         "seconds": [round(t, 4) for t in dt_all], "max_over_ranks_s": round(dt_max, 4),
                                         "host_enqueue_us_per_step": host_us_all, "host_thread_cpu_us_per_step": host_cpu_all,
                                         "host_note": "enqueue = wall time of one inference_batched() call on an idle GPU (all launches of a step queued, no "
                                                      "device->host read): a rank is host-bound when this approaches ms_per_step; thread_cpu = CPU time of the "
                                                      "launching thread over the timed steps (includes the spin of the step's one blocking read)"},
            "roofline": roofline, "value_inference_batched": through_forward, "pipelined": pipelined, "graphed": graphed,
            "eval_loop_host_inputs": eval_loop, "value_inputs": "device-resident (the 8 fp32 images are in HBM before the timed region, as the bench contract asks; `eval_loop_host_inputs` is the PCIe-inclusive loop)",
            "cpu_baseline": cpu_baseline, "timed_batch_parity": parity,
        }
        line.update(extras)
        print(json.dumps(line))
        if parity is not None and not (parity["detection_counts_equal"] and parity["gate_ok"]):
            sys.stdout.flush()
            sys.stderr.write("bench.py: the timed batch's detections do not match the CPU oracle: %s\n" % json.dumps(parity))
            os._exit(4)
        r101p = (extras.get("train", {}).get("r101_inference") or {}).get("parity")
        if r101p is not None and not r101p["gate_ok"]:
            sys.stdout.flush()
            sys.stderr.write("bench.py: the R101 leg's detections do not match the CPU oracle: %s\n" % json.dumps(r101p))
            os._exit(4)


def _live_conv_pmc(timeout_s=90):
    """HBM-side traffic and matrix-pipe occupancy of the dominant kernel's largest launch (3x3 256 -> 256 on the batch's p2 map:
    fpn_output2 / rpn_head.conv), measured on THIS box: three rocprofv3 passes (--pmc FETCH_SIZE | --pmc WRITE_SIZE | the SQ
    counters; each with --kernel-trace only, as MI355X_MICROARCH.md prescribes) of scripts/probe_one.py in a subprocess.
    Returns None when rocprofv3 is not there or a pass fails (the committed profile is quoted instead)."""
    import csv
    import glob
    import shutil
    import tempfile

    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    probe = os.path.join(ROOT, "scripts", "probe_one.py")
    if not os.path.exists(exe) or not os.path.exists(probe):
        return None
    tmp = tempfile.mkdtemp(prefix="lvc_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    agg, ms, kname = {}, None, "?"
    try:
        passes = ["FETCH_SIZE", "WRITE_SIZE", "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES"]
        for i, ctr in enumerate(passes):
            cmd = [exe, "--pmc"] + ctr.split() + ["--kernel-trace", "--output-format", "csv", "-d", tmp, "-o", "t%d" % i, "--",
                                                  sys.executable, probe, "8", "256", "200", "336", "256", "3", "1", "1"]
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s)
            if r.returncode != 0:
                return None
            for f in glob.glob(os.path.join(tmp, "**", "t%d*counter_collection.csv" % i), recursive=True):
                for row in csv.DictReader(open(f)):
                    if "conv3x3" in row["Kernel_Name"]:
                        agg[row["Counter_Name"]] = float(row["Counter_Value"])      # the last launch of the pass
                        kname = row["Kernel_Name"].split("(")[0]
            if i == 2:
                for f in glob.glob(os.path.join(tmp, "**", "t2*kernel_trace.csv"), recursive=True):
                    d = [(int(x["End_Timestamp"]) - int(x["Start_Timestamp"])) / 1e6 for x in csv.DictReader(open(f)) if "conv3x3" in x["Kernel_Name"]]
                    ms = d[-1] if d else None
        if "FETCH_SIZE" not in agg or "WRITE_SIZE" not in agg:
            return None
        cyc = agg.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        alg = 8 * 200 * 336 * 256 * 4 * 2 + 256 * 2304 * 4      # the p2 activation tensor read once and written once + the weights
        out = {"source": "live: rocprofv3 --pmc passes of scripts/probe_one.py 8 256 200 336 256 3 1 1 on this box, inside this bench run",
               "launch": "the 3x3 layer 256 -> 256 on [8,200,336,256] (fpn_output2; the RPN head on p2 is the same shape) on the kernel the detector runs it on: " + kname,
               "hbm_bytes_per_launch": int(2 * agg["FETCH_SIZE"] * 1024 + agg["WRITE_SIZE"] * 1024),
               "correction": "FETCH_SIZE doubled (gfx950 tallies the 128-byte requests of wide coalesced reads at 64 B: MI355X_MICROARCH.md, HBM), WRITE_SIZE as is",
               "algorithmic_bytes_per_launch": alg,
               "traffic_over_algorithmic": round((2 * agg["FETCH_SIZE"] * 1024 + agg["WRITE_SIZE"] * 1024) / alg, 3),
               "launch_ms_under_profiler": round(ms, 4) if ms else None,
               "mfma_busy_fraction": round(agg.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (cyc * 1024), 4) if cyc else None,
               "clock_GHz_under_load": round(cyc / ms / 1e6, 3) if (cyc and ms) else None}
        return out
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def _live_step_pmc(timeout_s=150):
    """Matrix-pipe occupancy of the WHOLE backbone and fabric traffic of the WHOLE step, measured on this box: three rocprofv3
    passes (--pmc FETCH_SIZE | --pmc WRITE_SIZE | --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES; each with --kernel-trace only, as
    MI355X_MICROARCH.md prescribes) of scripts/probe_step_pmc.py, which runs one step and one backbone-only pass between marker
    launches.  Busy fraction = sum of SQ_VALU_MFMA_BUSY_CYCLES / sum of (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs) over the launches,
    i.e. weighted by each launch's duration.  None when rocprofv3 is missing or a pass fails."""
    import csv
    import glob
    import shutil
    import tempfile

    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    probe = os.path.join(ROOT, "scripts", "probe_step_pmc.py")
    if not os.path.exists(exe) or not os.path.exists(probe):
        return None
    tmp = tempfile.mkdtemp(prefix="lvc_step_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    try:
        seg = {}       # counter -> [rows of the step, rows of the backbone pass]; a row = (kernel name, value)
        for i, ctr in enumerate(["FETCH_SIZE", "WRITE_SIZE", "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES"]):
            cmd = [exe, "--pmc"] + ctr.split() + ["--kernel-trace", "--output-format", "csv", "-d", tmp, "-o", "s%d" % i, "--", sys.executable, probe, str(BATCH_PER_GPU)]
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s)
            if r.returncode != 0:
                return None
            rows = []
            for f in glob.glob(os.path.join(tmp, "**", "s%d*counter_collection.csv" % i), recursive=True):
                rows += list(csv.DictReader(open(f)))
            rows.sort(key=lambda x: int(x["Dispatch_Id"]))
            for name in ctr.split():
                mine = [x for x in rows if x["Counter_Name"] == name]
                marks = [j for j, x in enumerate(mine) if "gelu" in x["Kernel_Name"]]
                if len(marks) < 3:
                    return None
                a, b, e = marks[-3], marks[-2], marks[-1]
                seg[name] = [[(x["Kernel_Name"].split("(")[0], float(x["Counter_Value"])) for x in mine[a + 1:b]],
                             [(x["Kernel_Name"].split("(")[0], float(x["Counter_Value"])) for x in mine[b + 1:e]]]
        step_bytes = 2 * 1024 * sum(v for _, v in seg["FETCH_SIZE"][0]) + 1024 * sum(v for _, v in seg["WRITE_SIZE"][0])
        bb_bytes = 2 * 1024 * sum(v for _, v in seg["FETCH_SIZE"][1]) + 1024 * sum(v for _, v in seg["WRITE_SIZE"][1])

        def busy(rows_busy, rows_act, pred):
            bsum = sum(v for n, v in rows_busy if pred(n))
            asum = sum(v for n, v in rows_act if pred(n)) / 8.0 * 1024.0
            return (bsum / asum) if asum else None

        conv = lambda n: any(t in n for t in ("conv", "stem_pool", "gemm"))
        per = {}
        for n, v in seg["GRBM_GUI_ACTIVE"][1]:
            per.setdefault(n, [0.0, 0.0, 0])
            per[n][1] += v / 8.0 * 1024.0
            per[n][2] += 1
        for n, v in seg["SQ_VALU_MFMA_BUSY_CYCLES"][1]:
            per.setdefault(n, [0.0, 0.0, 0])[0] += v
        tot_act = sum(v[1] for v in per.values())
        return {"source": "live: three rocprofv3 --pmc passes of scripts/probe_step_pmc.py (one step, one backbone-only pass, batch %d) on this box, inside this bench run" % BATCH_PER_GPU,
                "backbone_mfma_busy": round(busy(seg["SQ_VALU_MFMA_BUSY_CYCLES"][1], seg["GRBM_GUI_ACTIVE"][1], lambda n: True), 4),
                "backbone_mfma_busy_conv_kernels_only": round(busy(seg["SQ_VALU_MFMA_BUSY_CYCLES"][1], seg["GRBM_GUI_ACTIVE"][1], conv), 4),
                "whole_step_mfma_busy": round(busy(seg["SQ_VALU_MFMA_BUSY_CYCLES"][0], seg["GRBM_GUI_ACTIVE"][0], lambda n: True), 4),
                "backbone_launches": len(seg["GRBM_GUI_ACTIVE"][1]), "step_launches": len(seg["GRBM_GUI_ACTIVE"][0]),
                "backbone_by_kernel": {n: {"launches": v[2], "share_of_backbone_cycles": round(v[1] / tot_act, 4), "mfma_busy": round(v[0] / v[1], 4) if v[1] else None}
                                       for n, v in sorted(per.items(), key=lambda kv: -kv[1][1])[:8]},
                "step_hbm_bytes": int(step_bytes), "backbone_hbm_bytes": int(bb_bytes),
                "correction": "FETCH_SIZE x 2 (gfx950 tallies the 128-byte requests of wide coalesced reads at 64 B: MI355X_MICROARCH.md, HBM) + WRITE_SIZE, KiB -> bytes",
                "weighting": "sum of busy cycles / sum of active SIMD cycles over the launches (each launch weighs its own duration under the profiler)"}
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def _knn_kernels_alone(c):
    """The kernels of the kNN sweep alone, event-timed: the fp16 pre-filter GEMM on the full 120k x 2400 x 1024 problem, the
    verification kernel on its output (class-structured descriptors as in `knn_leg`), and the single-stage top-k kernel on a
    32768 x 2400 block."""
    import torch

    from lvc_amd import kernels as K
    from lvc_amd import label_verification as LV

    out = {}
    cls = torch.arange(80, device=c.dev).repeat_interleave(30)
    g = torch.Generator(device=c.dev).manual_seed(5)
    centers = torch.randn(80, KNN_D, device=c.dev, generator=g)
    qcls = torch.randint(0, 80, (KNN_Q,), device=c.dev, generator=g)
    shots = _knn_inputs(c, KNN_S, 7, cls, centers, 2.0)
    q = _knn_inputs(c, KNN_Q, 100, qcls, centers, 2.5)
    mu = K.colmean(shots)
    sn, sh, _, sres = K.rownorm_h(shots, mu=mu, eps=1e-8, mode=1, want_resid=True)
    _, qh, den, qres = K.rownorm_h(q, mu=mu, eps=1e-8, mode=1, want_rows=False, want_resid=True)
    sres_max = K.max_f32(sres)
    nq = min(KNN_Q, (2 ** 31 - 1) // (4 * KNN_S))
    q15 = bool(LV.KNN_Q15)       # what the sweep runs: the similarity matrix as 16-bit fixed point between the two stages
    eb = 2 if q15 else 4
    ms = _event_ms(lambda: K.gemm_f16(qh[:nq], sh, q15=q15), 10)
    fl = 2.0 * nq * KNN_S * KNN_D
    out["knn_gemm_f16"] = {"kernel": "gemm_f16_dma_kernel, %d x %d x %d fp16 operands -> %s" % (nq, KNN_S, KNN_D, "int16 (q15)" if q15 else "fp32"), "ms": round(ms, 4),
                           "tflops": round(fl / ms / 1e9, 1), "frac_of_fp16_mfma_peak_2500": round(fl / ms / 1e9 / 2500.0, 4),
                           "algorithmic_bytes": nq * KNN_D * 2 + KNN_S * KNN_D * 2 + nq * KNN_S * eb}
    out["knn_gemm_f16"]["GBps"] = round(out["knn_gemm_f16"]["algorithmic_bytes"] / ms / 1e6, 1)
    ap = K.gemm_f16(qh[:nq], sh, q15=q15)
    mg = LV.VERIFY_MARGIN + (LV.Q15_MARGIN if q15 else 0.0)
    # the per-row margins the sweep hands in (label_verification.knn_sweep)
    margins = (LV.pre_filter_margins(qres, 1.0 + 1e-6 + sres_max, sres_max, KNN_D) + (LV.Q15_MARGIN if q15 else 0.0))[:nq].contiguous()
    ms = _event_ms(lambda: K.knn_verify_topk_vote(ap, q[:nq], sn, mg, cls, qcls[:nq], 10, mu=mu, den=den, margins=margins), 10)
    alg = nq * KNN_S * eb + nq * KNN_D * 4 + nq * 10 * 8 + nq * 8
    out["knn_verify_topk_vote"] = {"kernel": "%s, %d x %d pre-filter similarities + the raw query rows -> exact top-10 class ids + keep" % (
                                       "knn_verify_q15_kernel" if q15 else "knn_verify_topk_vote_kernel", nq, KNN_S),
                                   "ms": round(ms, 4), "algorithmic_bytes": alg, "GBps": round(alg / ms / 1e6, 1),
                                   "frac_of_hbm_peak": round(alg / ms / 1e6 / PEAK_HBM_GBPS, 4)}
    ms = _event_ms(lambda: K.rownorm_h(q, mu=mu, eps=1e-8, mode=1, want_rows=False), 10)
    alg = KNN_Q * KNN_D * 6 + KNN_Q * 4
    out["knn_rownorm_h"] = {"kernel": "rownorm_h_kernel, %d x %d fp32 -> fp16 rows + denominators" % (KNN_Q, KNN_D), "ms": round(ms, 4),
                            "algorithmic_bytes": alg, "GBps": round(alg / ms / 1e6, 1), "frac_of_hbm_peak": round(alg / ms / 1e6 / PEAK_HBM_GBPS, 4)}
    del ap
    Q = 32768
    sims = torch.randn(Q, KNN_S, device=c.dev)
    det = torch.randint(0, 80, (Q,), device=c.dev)
    ms = _event_ms(lambda: K.knn_topk_vote(sims, KNN_S, cls, det, 10), 10)
    alg = Q * KNN_S * 4 + Q * 10 * 8 + Q * 8
    out["knn_topk_vote"] = {"kernel": "knn_topk_vote_kernel (single-stage path), %d x %d similarities -> top-10 class ids + keep" % (Q, KNN_S),
                            "ms": round(ms, 4), "algorithmic_bytes": alg, "GBps": round(alg / ms / 1e6, 1),
                            "frac_of_hbm_peak": round(alg / ms / 1e6 / PEAK_HBM_GBPS, 4)}
    return out


def _match(boxes, scores, classes, gboxes, gscores, gclasses, box_tol, score_tol):
    """Greedy one-to-one matching of oracle detections to GPU detections of the same class within the tolerances:
    (matched count, worst |box| error, worst |score| error among the matched)."""
    used, matched, wb, ws = set(), 0, 0.0, 0.0
    for i in range(len(gboxes)):
        db = (boxes - gboxes[i]).abs().max(dim=1)[0]
        ds = (scores - gscores[i]).abs()
        ok = (db <= box_tol) & (ds <= score_tol) & (classes == gclasses[i])
        for u in used:
            ok[u] = False
        idx = ok.nonzero().view(-1)
        if len(idx):
            j = int(idx[db[idx].argmin()])
            used.add(j)
            matched += 1
            wb, ws = max(wb, float(db[j])), max(ws, float(ds[j]))
    return matched, wb, ws


def range_rehearsal(c, steps, warmup):
    """a checkpoint-shaped rehearsal of the fp16 forms' range envelope.  No trained weights exist in this environment; the
    detector is built with an MSRA-initialised trunk behind identity FrozenBN, loaded through the pre-v3 path (no running statistics:
    lvc_amd.utils.synthetic.msra_checkpoint_rehearsal_), so that activations reach O(1e3..1e4) in the deep stages.  Reported: how many
    layers leave the one-accumulator forms (|a| <= 4094), where they go, how many passes that takes, and the steady-state rate with
    them re-routed (the bench batch: 8 x 800x1333)."""
    import torch

    from lvc_amd import kernels as K
    from lvc_amd.config.presets import base_rcnn_fpn
    from lvc_amd.layers.wrappers import Conv2d, Linear
    from lvc_amd.modeling import build_model
    from lvc_amd.modeling.backbone.resnet import BottleneckBlock
    from lvc_amd.utils import synthetic as syn

    model = build_model(base_rcnn_fpn(device="cuda:%d" % c.local_rank)).eval()
    syn.msra_checkpoint_rehearsal_(model)
    batch = [{"image": syn.synthetic_image(1 + i).to(c.dev), "height": 800, "width": 1333} for i in range(BATCH_PER_GPU)]
    epoch0, split0 = K.RANGE_EPOCH, K.CONV_SPLIT
    with torch.no_grad():
        feats = model.backbone.bottom_up(model.preprocess_image(batch).tensor)      # (re-routes nothing by itself: the words are read by forward())
        scales = {k: round(float(v.abs().max()), 1) for k, v in feats.items()}
        del feats
        K.clear_conv_error_word(c.dev)
        passes = 0
        while passes < 64:      # forward() repeats a flagged pass itself (8 tries); the loop covers a longer cascade
            try:
                model(batch)
                break
            except K.Fp16RangeError:
                passes += 1
        for _ in range(warmup):
            model(batch)
        epoch1 = K.RANGE_EPOCH
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = model(batch)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    tiers = {n: m._range_state["tier"] for n, m in model.named_modules() if isinstance(m, (Conv2d, Linear)) and m._range_state["tier"]}
    layers = sum(1 for _n, m in model.named_modules() if isinstance(m, (Conv2d, Linear)))
    blocks = [m for m in model.modules() if isinstance(m, BottleneckBlock)]
    res = {"workload": "R50-FPN inference, bs=%d synthetic 3x800x1333, MSRA-initialised trunk behind identity FrozenBN loaded through the pre-v3 "
                       "state_dict path (no trained checkpoint exists here)" % BATCH_PER_GPU,
           "activation_scale_max": scales,
           "layers_total": layers, "layers_off_the_one_accumulator_form": len(tiers),
           "to_two_accumulators": sum(1 for t in tiers.values() if t == 1), "to_bf16x3": sum(1 for t in tiers.values() if t == 2),
           "fused_blocks_off": sum(1 for b in blocks if getattr(b, "_bneck_state", {}).get("off")),
           "chained_pairs_off": sum(1 for b in blocks if getattr(b, "_chain_state", {}).get("off")),
           "rerouting_passes": K.RANGE_EPOCH - epoch0, "settled": K.RANGE_EPOCH == epoch1,
           "value": round(BATCH_PER_GPU * steps / dt, 2), "unit": "img/s", "ms_per_step": round(1e3 * dt / steps, 3),
           "detections_per_image": [len(o["instances"]) for o in out],
           "rerouted_layers": {k: ("two accumulators" if t == 1 else "bf16x3") for k, t in sorted(tiers.items())}}
    res["process_wide_split_after"] = K.CONV_SPLIT      # "f16x2" unless an operand without a per-layer range word overflowed (stem, weights)
    K.CONV_SPLIT = split0                                 # the legs after this one run as configured
    del model
    torch.cuda.empty_cache()
    return res


def _gpu_proposals(model, batch):
    """The proposal stage's output for `batch` (what inference_batched hands to the heads), per image [n,4] on the CPU."""
    import torch

    with torch.no_grad():
        images = model.preprocess_image(batch)
        sizes_dev = model._dev_const(images.image_sizes, torch.int32)
        N, _, Hp, Wp = images.tensor.shape
        x4 = images.tensor.as_strided((N, Hp, Wp, 4), (Hp * Wp * 4, Wp * 4, 4, 1), images.tensor.storage_offset())
        pb, _pl, pc = model.proposal_generator.predict_proposals_batched(model.backbone.forward_nhwc(x4), sizes_dev)
    pc = pc.tolist()
    return [pb[i, : pc[i]].cpu() for i in range(len(pc))]


def _cpu_baseline(model, imgs, gpu_dets, gpu_props=None):
    """the CPU restatement (oracle/) on this host, 5 warm-up + 20 timed single-image forwards
    (bounded to ~60 s of CPU time: fewer timed iterations are taken, and reported, on a slower host).  The forwards cycle
    through the images of the TIMED GPU batch and their detections are kept: `timed_batch_parity` compares them with what the
    GPU produced in the last timed step (`gpu_dets`: per image (boxes, scores, classes) on the CPU)."""
    import torch

    from oracle import rcnn as orc

    # torch-CPU convolutions stop scaling (and regress) far below the 256 hardware threads of the GPU host, so the
    # baseline uses 32 threads -- stated as `cores`
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    cpu_in = [[{"image": im.cpu(), "height": 800, "width": 1333}] for im in imgs]
    spec = orc.RCNNSpec()
    ref = {}
    with torch.no_grad():
        t1 = time.perf_counter()
        ref[0] = orc.generalized_rcnn_inference(sd, spec, cpu_in[0])[0]
        first = time.perf_counter() - t1
        nwarm = 1
        while nwarm < 5 and (time.perf_counter() - t1) < 12.0:
            ref[nwarm % len(imgs)] = orc.generalized_rcnn_inference(sd, spec, cpu_in[nwarm % len(imgs)])[0]
            nwarm += 1
        n, t1 = 0, time.perf_counter()
        while n < 20 and (n < 3 or (time.perf_counter() - t1) < 45.0):
            i = (nwarm + n) % len(imgs)
            ref[i] = orc.generalized_rcnn_inference(sd, spec, cpu_in[i])[0]
            n += 1
        cdt = time.perf_counter() - t1
    # the bars: K_NOISE (= 2) x the CPU path's OWN rounding noise (fp32 vs fp64 evaluation of the same weights) on two of the batch's images
    from oracle import noise as onoise

    with torch.no_grad():
        t2 = time.perf_counter()
        nz_imgs = sorted(ref)[:2]
        props32 = None
        if gpu_props is not None:      # the fp32 oracle's proposals on the noise images (one more fp32 pass each: outside the timed loop)
            props32 = [orc.generalized_rcnn_inference(sd, spec, cpu_in[i], return_intermediates=True)[1]["proposals"][0][0] for i in nz_imgs]
        nz = onoise.fp32_vs_fp64(sd, spec, [cpu_in[i][0] for i in nz_imgs], res32=[ref[i] for i in nz_imgs], props32=props32)
        nz_s = time.perf_counter() - t2
    dev = onoise.deviation([gpu_dets[i] for i in sorted(ref)], [(ref[i]["pred_boxes"], ref[i]["scores"], ref[i]["pred_classes"]) for i in sorted(ref)])
    dev_same = onoise.deviation([gpu_dets[i] for i in nz_imgs], [(ref[i]["pred_boxes"], ref[i]["scores"], ref[i]["pred_classes"]) for i in nz_imgs])
    prop_dev = onoise.proposal_deviation([gpu_props[i] for i in nz_imgs], props32) if props32 is not None else None
    gate_ok, bars, gate_msg = onoise.gate(dev, nz, dev_same=dev_same, prop_dev=prop_dev)
    # parity of the timed batch: every image the oracle saw
    tot = loose = tight = 0
    wb = ws = 0.0
    counts_equal = True
    for i, r in sorted(ref.items()):
        gb, gs, gc = gpu_dets[i]
        counts_equal &= len(gs) == len(r["scores"])
        m, b_, s_ = _match(gb, gs, gc, r["pred_boxes"], r["scores"], r["pred_classes"], 0.1, 2e-3)
        t, _, _ = _match(gb, gs, gc, r["pred_boxes"], r["scores"], r["pred_classes"], 1e-3, 1e-3)
        tot += len(r["scores"]); loose += m; tight += t
        wb, ws = max(wb, b_), max(ws, s_)
    parity = {"images_checked": sorted(ref), "oracle_detections": tot, "detection_counts_equal": bool(counts_equal),
              "matched_fraction_0.1px_2e-3": round(loose / max(1, tot), 4), "matched_fraction_1e-3": round(tight / max(1, tot), 4),
              "worst_box_px_among_matched": round(wb, 5), "worst_score_among_matched": round(ws, 6),
              "deviation_among_matched": {k: (round(v, 6) if isinstance(v, float) else v) for k, v in dev.items()},
              "cpu_path_fp32_vs_fp64_noise": dict({k: (round(v, 6) if isinstance(v, float) else v) for k, v in nz.items()},
                                                  images=nz_imgs, seconds=round(nz_s, 1)),
              "bars": {k: round(v, 6) for k, v in bars.items()}, "gate": gate_msg, "gate_ok": bool(gate_ok),
              "deviation_on_the_noise_images": {k: (round(v, 6) if isinstance(v, float) else v) for k, v in dev_same.items()},
              "proposal_stage_on_the_noise_images": prop_dev,
              "pass_bar": "every bar is measured in this run on the CPU path itself (oracle/noise.py; `bars`): equal counts; matched_fraction >= the fraction "
                          "the CPU path finds of its OWN fp64 detections - %g (identity); median / p90 of the matched |box|, |score| differences <= %g x its "
                          "fp32-vs-fp64 noise; the largest matched difference on the noise images <= %g x its largest; one stage earlier, the fraction of "
                          "the CPU path's PROPOSALS found within 0.1 px >= its own fp32-vs-fp64 fraction - %g and the proposal counts no further apart than "
                          "2 x its own + 2; else the run exits non-zero"
                          % (onoise.IDENT_MARGIN, onoise.K_NOISE, onoise.K_MAX, onoise.IDENT_MARGIN),
              "oracle_pinning": "oracle/rcnn.py is pinned against the imported reference's outputs (tests/golden, tests/test_oracle_golden.py) -- except "
                                "torchvision's NMS (absent from the reference tree; independent witness: the reference's rotated-NMS kernel) and the DINO "
                                "ViT of the descriptor leg (weights are network-only): those two oracles are unpinned",
              "note": "GPU detections of the LAST TIMED step vs oracle/rcnn.py (fp32 CPU) on the same images; both are fp32 evaluations "
                      "of a 53-layer trunk, each ~2e-3 px (median) from the fp64 answer (tests/test_gpu_chain.py), so the literal 1e-3 "
                      "fraction is what two valid fp32 paths share"}
    return {"value": round(n / cdt, 4), "unit": "img/s", "s_per_img": round(cdt / n, 4), "cores": cores,
            "host_threads": os.cpu_count(), "cpu_model": _cpu_model(), "kind": "port",
            "sample": "%d warm-up + %d timed single-image forwards (bs=1, 3x800x1333) cycling over the timed batch's %d images, through "
                      "oracle/rcnn.py (torch-CPU convs, oracle.c ROIAlign/NMS); first call %.2f s" % (nwarm, n, len(imgs), first)}, parity


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=("infer", "train", "knn"), default="infer")
    ap.add_argument("--knn-inputs", choices=("randn", "structured"), default=None,
                    help="kNN leg: time one kind of query / shot rows only (profiler traces); default both")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the bandwidth_kernels / knn / dp_legs objects")
    ap.add_argument("--no-live-pmc", action="store_true", help="skip the rocprofv3 --pmc subprocess passes (the committed profile is quoted)")
    ap.add_argument("--no-launch-timer", action="store_true", help="skip the per-launch HIP events (A/B their cost)")
    ap.add_argument("--pipeline-depth", type=int, default=2,
                    help="streams of the extra pipelined pass reported as `pipelined` (1 = skip it); the timed region is always one stream")
    args = ap.parse_args()
    global KNN_INPUTS
    KNN_INPUTS = args.knn_inputs
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        _self_launch(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print(json.dumps({"error": "--gpus %d but the launcher started %d rank(s)" % (args.gpus, world)}))
        sys.exit(2)
    c = _setup()
    import torch.distributed as dist

    if args.workload == "infer":
        infer_main(c, args)
    else:
        leg = (train_leg if args.workload == "train" else knn_leg)(c, args.steps, args.warmup)
        if c.rank == 0:
            if args.workload == "train":
                line = {"metric": "img/s cfg-3 novel fine-tune step, R50-FPN 800x1333, data parallel over RCCL", "value": leg["value"],
                        "unit": "img/s", "scaling": "weak", "dtype": "f32"}
            else:
                line = {"metric": "queries/s label-verification kNN sweep (120k x 2400 x 1024)", "value": leg["queries_per_s"],
                        "unit": "queries/s", "scaling": "strong", "dtype": "f32"}
            line.update({"n_gpus": c.world, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True, "vs_baseline": None,
                         "ms_per_step": leg.get("ms_per_step", leg.get("ms_per_sweep")),
                         "data": "synthetic", "config": {"workload": leg["workload"], "parallelism": "dp%d" % c.world},
                         "rccl": c.rccl, "detail": leg})
            print(json.dumps(line))
    if c.use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
