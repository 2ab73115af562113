"""Worker of tests/test_gpu_dist.py: one rank of a 2-rank data-parallel check (launched by torch.distributed.run).

  mode "train": cfg-3 fine-tune gradients.  Every rank runs forward + backward on ITS two images and the gradients are
                averaged with lvc_amd.distributed.allreduce_gradients_; rank 0 then runs the concatenated 4-image batch
                alone and the averaged gradient must equal that single-process gradient (DDP's contract,
                reference lvc/engine/defaults.py:326-331).  torch.randperm is the identity on every side so the
                sampled RoIs are defined by position.
  mode "knn":   knn_sweep_distributed (shots all-gathered, queries sharded, results gathered to rank 0) must equal the
                single-process sweep over all queries.
Backend: RCCL ("nccl") when every rank has its own GPU, otherwise gloo with both ranks on cuda:0 (same code path above
the collective, device tensors through gloo)."""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    mode = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    own_gpu = torch.cuda.device_count() >= world
    dev = torch.device("cuda", rank if own_gpu else 0)
    torch.cuda.set_device(dev)
    backend = "nccl" if own_gpu else "gloo"
    dist.init_process_group(backend=backend)
    from lvc_amd import distributed as D

    out = {"backend": backend, "world": dist.get_world_size()}
    if mode == "train":
        from lvc_amd.config import set_global_cfg
        from lvc_amd.config.presets import base_rcnn_fpn
        from lvc_amd.modeling import build_model
        from lvc_amd.structures import Boxes, Instances
        from lvc_amd.utils import synthetic as syn
        from lvc_amd.utils.events import EventStorage

        torch.randperm = lambda n, **kw: torch.arange(n, **{k: v for k, v in kw.items() if k in ("device", "dtype")})
        cfg = base_rcnn_fpn(num_classes=20, device=str(dev))
        cfg.MODEL.BACKBONE.FREEZE = True
        cfg.MODEL.PROPOSAL_GENERATOR.FREEZE = True
        cfg.MODEL.ROI_HEADS.FREEZE_FEAT = True
        set_global_cfg(cfg)
        model = build_model(cfg)
        syn.conditioned_r50_fpn_(model)
        model.train()

        def image(i):
            g = torch.Generator().manual_seed(50 + i)
            h, w, n = 320, 480, 6
            x1 = torch.rand(n, generator=g) * (w - 120)
            y1 = torch.rand(n, generator=g) * (h - 120)
            inst = Instances((h, w))
            inst.gt_boxes = Boxes(torch.stack([x1, y1, x1 + 30 + torch.rand(n, generator=g) * 80,
                                               y1 + 30 + torch.rand(n, generator=g) * 80], 1))
            inst.gt_classes = torch.randint(0, 20, (n,), generator=g)
            return {"image": syn.synthetic_image(3 + i, h, w).to(dev), "instances": inst, "height": h, "width": w}

        per = 2
        params = [p for p in model.parameters() if p.requires_grad]

        def grads(batch):
            for p in params:
                p.grad = None
            with EventStorage(0):
                losses = model(batch)
                sum(losses.values()).backward()
            return losses

        grads([image(rank * per + i) for i in range(per)])
        nbytes = D.allreduce_gradients_(params)
        avg = [p.grad.detach().clone() for p in params]
        out["bytes"] = nbytes
        # the bucketed, backward-overlapped exchange (cfg 5's path) in both forms must give the same averaged gradient
        for mode in ("all_reduce", "rs_ag"):
            buckets = D.GradientBuckets(params, bucket_bytes=1 << 18, mode=mode)   # several buckets
            grads([image(rank * per + i) for i in range(per)])
            buckets.finish()
            out["buckets_" + mode] = max(float((a - p.grad).abs().max() / a.abs().max()) for a, p in zip(avg, params))
            buckets.remove()
        if rank == 0:
            grads([image(i) for i in range(per * world)])
            rel = [float((a - p.grad).norm() / p.grad.norm()) for a, p in zip(avg, params)]
            out["rel_err_vs_concatenated_batch"] = rel
    else:
        from lvc_amd.label_verification import knn_sweep, knn_sweep_distributed

        g = torch.Generator().manual_seed(0)
        S, Dm, Q = 240, 128, 4096
        shots = torch.randn(S, Dm, generator=g)
        classes = torch.arange(24).repeat_interleave(10)
        perm = torch.randperm(S, generator=g)          # ranks hold unsorted, interleaved shots
        shots, classes = shots[perm], classes[perm]
        q = torch.randn(Q, Dm, generator=g)
        det = torch.randint(0, 24, (Q,), generator=g)
        sr, qr = D.shard_range(S, rank, world), D.shard_range(Q, rank, world)
        top, keep = knn_sweep_distributed(classes[sr.start: sr.stop].to(dev), shots[sr.start: sr.stop].to(dev),
                                          q[qr.start: qr.stop].to(dev), det[qr.start: qr.stop].to(dev), 10, True)
        if rank == 0:
            order = classes.argsort(stable=True)
            rt, rk = knn_sweep(classes[order].to(dev), shots[order].to(dev), q.to(dev), det.to(dev), 10, True)
            out["top_equal"] = bool(torch.equal(top, rt))
            out["keep_equal"] = bool(torch.equal(keep, rk))
            out["rows"] = int(top.shape[0])
    torch.cuda.synchronize()
    dist.barrier()
    if rank == 0:
        print("DP_WORKER_RESULT " + json.dumps(out), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
