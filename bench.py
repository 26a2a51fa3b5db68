#!/usr/bin/env python3
"""Headline benchmark: B-scans/s of one TRAINING step of the 256x512x1, 3-class U-Net (fwd + Dice loss +
bwd + gradient all-reduce + Adam), per-rank batch 32 (BASELINE.json configs[1]; configs[3] at N>1: weak
scaling, global batch 32*N), fp32, synthetic scans, random-init weights.  Also reports inference ms/B-scan.

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

Prints ONE JSON line on rank 0 (contract in the task brief) with `roofline` (dominant kernel, measured
live with HIP events on the launch stream by the library's own per-launch profiler) and `cpu_baseline`
(the torch-CPU port of the reference path from oracle/, timed on this box's host cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

# peaks from /opt/skills/guides/MI355X_MICROARCH.md (chip table)
PEAK_HBM_GBS = 8000.0        # HBM3E spec; 6290 measured by a float4 copy
PEAK_F32_TFLOPS = 157.3      # fp32 MFMA = fp32 vector peak (dense)


def cpu_baseline(H, W, C, budget_s=20.0):
    """Train-step throughput of the CPU port (oracle/unet_torch.py: torch-CPU/oneDNN restatement of the Keras
    graph, autograd, Keras-formula Adam) on this box's host cores.  Bounded sample: batch 4, >=2 timed steps."""
    from oracle import unet_numpy as on
    from oracle import unet_torch as ot
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    ncpu = max(1, min(ncpu, int(os.environ.get("OCT_CPU_THREADS", "16"))))   # GPU-box share for one GPU is 16
    torch.set_num_threads(ncpu)
    B = 4
    cfg = on.UNetConfig(num_classes=C)
    params, state = on.init_params(cfg, seed=0, dtype=np.float32)
    tp, ts = ot.to_torch(params, state, dtype=torch.float32, requires_grad=True)
    images, labels = on.synth_scans(B, H, W, C, seed=1)
    x = torch.tensor(on.preprocess_u8(images, np.float32))
    y = torch.nn.functional.one_hot(torch.tensor(labels[..., 0].astype("int64")), C).float()
    mask = (torch.rand(B, H >> 4, W >> 4, 128) > 0.5).float()
    leaves = [v for p in tp for v in p.values()]
    m = [torch.zeros_like(v) for v in leaves]; v2 = [torch.zeros_like(v) for v in leaves]

    def step(t):
        probs = ot.forward(cfg, tp, ts, x, training=True, dropout_mask=mask)
        loss = ot.dice_loss(y, probs, macro=True)
        grads = torch.autograd.grad(loss, leaves)
        lr_t = 1e-3 * np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
        with torch.no_grad():
            for p_, g_, m_, v_ in zip(leaves, grads, m, v2):
                m_.mul_(0.9).add_(g_, alpha=0.1); v_.mul_(0.999).addcmul_(g_, g_, value=0.001)
                p_.sub_(lr_t * m_ / (v_.sqrt() + 1e-7))

    step(1)
    t0 = time.perf_counter(); n = 0
    while n < 2 or (time.perf_counter() - t0 < budget_s and n < 50):
        step(n + 2); n += 1
    dt = time.perf_counter() - t0
    with torch.no_grad():
        ti = time.perf_counter()
        for _ in range(3):
            ot.forward(cfg, tp, ts, x, training=False)
        inf_ms = (time.perf_counter() - ti) / (3 * B) * 1e3
    return {"value": round(B * n / dt, 3), "unit": "B-scans/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} train steps of batch {B} ({H}x{W}x1, C={C}, fp32) with oracle/unet_torch.py "
                      f"(torch-CPU oneDNN port of the Keras graph; TensorFlow 2.9 is not installable here)",
            "host_cpus": os.cpu_count(), "inference_ms_per_scan": round(inf_ms, 2)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch (weak scaling)")
    ap.add_argument("--height", type=int, default=256)
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--classes", type=int, default=3)
    ap.add_argument("--infer-batch", type=int, default=128)
    ap.add_argument("--act-dtype", choices=["f32", "bf16"], default="f32",
                    help="activation STORAGE type (arithmetic is f32 either way); the headline metric is f32")
    ap.add_argument("--pool-layers", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--dump-profile", default=None, help="write the per-(kernel, layer) launch table to this JSON file")
    args = ap.parse_args()

    from oct_image_segmentation_models_amd import parallel
    from oct_image_segmentation_models_amd.common.synthetic import make_scans
    from oct_image_segmentation_models_amd.engine import UNetEngine

    # OCT_BENCH_REHEARSAL=1: multi-process rehearsal on a ONE-GPU box (all ranks share cuda:0, gloo collective);
    # never used by the driver -- real runs are one rank per GPU over RCCL
    rehearsal = os.environ.get("OCT_BENCH_REHEARSAL") == "1"
    rank, local_rank, world = parallel.init("gloo" if rehearsal else "nccl")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    dev = torch.device("cuda", 0 if rehearsal else local_rank)
    torch.cuda.set_device(dev)
    B, H, W, C = args.batch, args.height, args.width, args.classes

    eng = UNetEngine(device=dev, input_channels=1, num_classes=C, image_height=H, image_width=W,
                     max_batch=max(B, args.infer_batch), training=True, seed=1000 + rank, init_seed=0,
                     pool_layers=args.pool_layers, dtype={"f32": "float32", "bf16": "bfloat16"}[args.act_dtype])
    parallel.broadcast_parameters(eng.params, eng.state)
    # a few distinct synthetic scans per rank, tiled to the batch (host generation is not part of the step)
    nd = min(B, 8)
    images, labels = make_scans(nd, H, W, C, seed=1234 + rank)
    reps = (B + nd - 1) // nd
    x = torch.from_numpy(np.tile(images, (reps, 1, 1, 1))[:B]).to(dev)
    lab = torch.from_numpy(np.tile(labels[..., 0], (reps, 1, 1))[:B].copy()).to(dev)

    def train_step():
        eng.forward(x, training=True, labels=lab, want_probs=False)
        loss4 = eng.loss_dice()
        eng.backward(lab, macro=True, loss_scale=1.0 / world)
        parallel.allreduce_gradients(eng.grads)
        eng.adam_step(lr=1e-3)
        return loss4

    for _ in range(args.warmup):
        train_step()
    parallel.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss4 = train_step()
    torch.cuda.synchronize(); parallel.barrier()
    dt = parallel.max_over_ranks(time.perf_counter() - t0, dev)
    scans_per_s = world * B * args.steps / dt
    # which BASELINE.json configuration this run is: [1] = the headline (256x512x1, P=4, fp32, batch 32);
    # [2] = 512x1024x1, P=5, bf16 storage, batch 64; anything else is labelled custom
    if (H, W, C, args.pool_layers, args.act_dtype, B) == (256, 512, 3, 4, "f32", 32):
        cfg_label = "configs[1]"
    elif (H, W, args.pool_layers, args.act_dtype, B) == (512, 1024, 5, "bf16", 64):
        cfg_label = "configs[2]"
    else:
        cfg_label = "custom (not a BASELINE configuration)"
    final_loss = float(loss4[0])

    # ---- inference: hipGraph-replayed forward (+argmax), batch 128, inputs resident ----
    IB = args.infer_batch
    xi = torch.from_numpy(np.tile(images, ((IB + nd - 1) // nd, 1, 1, 1))[:IB]).to(dev)
    eng.graph_capture(xi, want_probs=True, want_argmax=True)
    for _ in range(3):
        eng.graph_launch()
    torch.cuda.synchronize()
    ti = time.perf_counter()
    n_inf = 10
    for _ in range(n_inf):
        eng.graph_launch()
    torch.cuda.synchronize()
    infer_ms = parallel.max_over_ranks((time.perf_counter() - ti) / (n_inf * IB) * 1e3, dev)

    out = {
        "metric": f"B-scans/sec (train step), {H}x{W} {C}-class U-Net", "value": round(scans_per_s, 2),
        "unit": "B-scans/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32" if args.act_dtype == "f32" else "f32 math / bf16 activation storage",
        "data": "synthetic",
        "config": {"workload": f"{cfg_label}: train step (fwd+Dice-macro+bwd+allreduce+Adam), per-GPU batch {B}, "
                               f"{H}x{W}x1, {C}-class, pool_layers={args.pool_layers}, start_neurons=8, random-init weights",
                   "global_batch": B * world, "parallelism": f"dp{world}"},
        "inference_ms_per_scan": round(infer_ms, 5), "inference_batch": IB, "final_loss": round(final_loss, 5),
    }

    if rank == 0 and not args.no_profile:
        # per-launch HIP-event profile of 3 further (untimed) steps, on the launch stream
        eng.profile_begin()
        nprof = 3
        for _ in range(nprof):
            eng.forward(x, training=True, labels=lab, want_probs=False)
            eng.loss_dice(); eng.backward(lab, macro=True, loss_scale=1.0 / world)
        ents = eng.profile_end()
        if args.dump_profile:
            with open(args.dump_profile, "w") as f:
                json.dump({"steps": nprof, "entries": ents}, f, indent=1)
        by_k = {}
        for e in ents:
            k = by_k.setdefault(e["kernel"], dict(ms=0.0, flops=0.0, bytes=0.0, launches=0))
            k["ms"] += e["total_ms"]; k["flops"] += e["flops"]; k["bytes"] += e["bytes"]; k["launches"] += e["launches"]
        total_ms = sum(k["ms"] for k in by_k.values())
        def roof_of(k):
            """achieved = ALGORITHMIC flops (or bytes) of the kernel's launches / their summed HIP-event duration"""
            ai = k["flops"] / max(k["bytes"], 1.0)
            if ai > PEAK_F32_TFLOPS * 1e12 / (PEAK_HBM_GBS * 1e9):
                ach = k["flops"] / (k["ms"] * 1e-3) / 1e12
                return {"bound": "mfma", "achieved": round(ach, 3), "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(ach / PEAK_F32_TFLOPS, 4)}
            ach = k["bytes"] / (k["ms"] * 1e-3) / 1e9
            return {"bound": "hbm", "achieved": round(ach, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": round(ach / PEAK_HBM_GBS, 4)}

        name, dom = max(by_k.items(), key=lambda kv: kv[1]["ms"])
        avg_us = dom["ms"] / dom["launches"] * 1e3
        roof = roof_of(dom)
        # HBM traffic of that kernel from the committed rocprofv3 --pmc passes of this same command (profiles/), per launch
        traffic = None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))["kernels"]
            v = pmc.get(name)    # the profiler tags are the exact instantiation names rocprofv3 reports
            if v is not None:
                traffic = round(v["hbm_bytes_per_launch"], 0)
        except (OSError, KeyError, ValueError):
            traffic = None
        roof.update({"traffic": traffic, "kernel": name, "avg_launch_us": round(avg_us, 2),
                     "launches_per_step": dom["launches"] // nprof,
                     "share_of_step_kernel_time": round(dom["ms"] / total_ms, 4),
                     "algorithmic_flops_per_launch": dom["flops"] / dom["launches"],
                     "algorithmic_bytes_per_launch": dom["bytes"] / dom["launches"]})
        out["roofline"] = roof
        top = sorted(by_k.items(), key=lambda kv: -kv[1]["ms"])[:10]
        out["kernel_time_ms_per_step"] = {k: round(v["ms"] / nprof, 4) for k, v in top}
        out["roofline_top_kernels"] = {k: dict(roof_of(v), ms_per_step=round(v["ms"] / nprof, 4)) for k, v in top}
        # whole-step roofline context: SURVEY 8d algorithmic work per scan (10.27 GFLOP, 253 MB) vs step time
        out["step_vs_roofline"] = {"algorithmic_gflop_per_scan": 10.27, "algorithmic_mb_per_scan": 253.0,
                                   "roofline_us_per_scan": 73.2,
                                   "achieved_us_per_scan": round(dt / args.steps / B * 1e6, 2)}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(H, W, C)
    if rank == 0:
        if rehearsal:
            out["data"] = "synthetic (REHEARSAL: ranks share one GPU, gloo collective -- not a measurement)"
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
