#!/usr/bin/env python3
"""Headline benchmark: B-scans/s of one TRAINING step of the 256x512x1, 3-class U-Net (fwd + Dice loss +
bwd + gradient all-reduce + Adam), per-rank batch 32 (BASELINE.json configs[1]; configs[3] at N>1: weak
scaling, global batch 32*N), fp32, synthetic scans, random-init weights.  Also reports inference ms/B-scan
(hipGraph forward alone, and end to end with the host min-path post-process: BASELINE configs[4]).

    python bench.py --gpus N --steps K --warmup W

With N>1 and no WORLD_SIZE in the environment the script starts its own N ranks (``torch.distributed.run``,
one per GPU, RCCL) as a child process -- decided before anything touches the GPU -- and exits with its code.

Prints ONE JSON line on rank 0 (contract in the task brief) with
  * ``roofline``: the dominant SOURCE kernel (template instantiations grouped), measured live with HIP events on
    the launch stream by the library's own per-launch profiler, plus ``step_frac`` (whole step vs the SURVEY A.3
    roofline of THIS run's shape/dtype) and ``traffic_ratio`` (PMC bytes / algorithmic bytes per step, from the
    committed rocprofv3 --pmc passes named in ``traffic_source``);
  * ``cpu_baseline``: the torch-CPU port of the reference path from oracle/, timed on this box's host cores;
  * ``allreduce``: what RCCL itself reports (backend, world size, every rank's device) and the per-step cost of the two
    gradient all-reduces.  At N = 1 the timed steps run WITHOUT a collective (that is the single-GPU metric); a short
    extra leg then creates a ONE-rank RCCL communicator and drives the same exchange step through it
    (``--no-collective-leg`` skips it, ``--force-collective`` puts it inside the timed region instead);
  * ``fit``: B-scans/s of ``Model.fit`` fed by ``DataGenerator`` from a uint8 array set (host gather -> pinned double
    buffer -> H2D -> step), i.e. the real training loop beside the resident-input ``value``.
"""
import argparse
import glob
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# peaks from /opt/skills/guides/MI355X_MICROARCH.md (chip table)
PEAK_HBM_GBS = 8000.0        # HBM3E spec; 6290 measured by a float4 copy
MEASURED_HBM_GBS = 6290.0
PEAK_F32_TFLOPS = 157.3      # fp32 MFMA = fp32 vector peak (dense)
PEAK_BF16_TFLOPS = 2500.0    # dense bf16 MFMA


# ---------------------------------------------------------------------------------------------------------------
# SURVEY Appendix A.3 cost model, computed from the run's own shape (not constants)
# ---------------------------------------------------------------------------------------------------------------
def unet_layers(H, W, in_ch=1, C=3, sn=8, P=4, L=2):
    """(name, k, cin, cout, h, w, src) in Keras creation order (reference models/unet.py:106-153)."""
    out, cin = [], in_ch
    for i in range(P):
        for j in range(L):
            src = "input" if (i == 0 and j == 0) else ("pool" if j == 0 else "prev")
            out.append((f"enc{i}.conv{j}", 3, cin, sn << i, H >> i, W >> i, src)); cin = sn << i
    for j in range(L):
        out.append((f"mid.conv{j}", 3, cin, sn << P, H >> P, W >> P, "pool" if j == 0 else "prev")); cin = sn << P
    for i in range(P):
        lvl = P - 1 - i; size = sn << lvl
        out.append((f"dec{i}.up", 2, cin, size, H >> lvl, W >> lvl, "up")); cin = 2 * size
        for j in range(L):
            out.append((f"dec{i}.conv{j}", 3, cin, size, H >> lvl, W >> lvl, "concat" if j == 0 else "prev")); cin = size
    out.append(("head", 1, cin, C, H, W, "head"))
    return out


def cost_model(H, W, C=3, sn=8, P=4, L=2, es=4, peak_tflops=PEAK_F32_TFLOPS, bw_gbs=MEASURED_HBM_GBS):
    """Algorithmic work per B-scan: forward = every conv reads its logical input once (low-res tensor
    for an up-conv, both halves of a concat) and writes its output once, pooled tensors written once; backward = per
    conv dY + X read, dX written (none for the first conv), the saved z read once more for the BN-backward sums;
    roofline time = sum over layers of max(flops / peak, bytes / measured HBM rate).  cfg-A fp32 reproduces SURVEY's
    3.431 GF / 85.7 MB forward, 10.27 GF / 253 MB and 73.2 us per training step."""
    peak, bw = peak_tflops * 1e12, bw_gbs * 1e9
    f_fl = f_by = b_fl = b_by = t_f = t_b = 0.0
    for (_, k, cin, cout, h, w, src) in unet_layers(H, W, 1, C, sn, P, L):
        px = h * w
        fl = 2.0 * k * k * cin * cout * px
        inb = (px / 4 if src == "up" else px) * cin * es
        outb = px * cout * es
        f_fl += fl; f_by += inb + outb; t_f += max(fl / peak, (inb + outb) / bw)
        bfl = fl if src == "input" else 2 * fl
        bb = outb + inb + (0 if src == "input" else inb) + (0 if src == "head" else outb)
        b_fl += bfl; b_by += bb; t_b += max(bfl / peak, bb / bw)
    for i in range(P):
        pb = (H >> (i + 1)) * (W >> (i + 1)) * (sn << i) * es
        f_by += pb; b_by += pb; t_f += pb / bw; t_b += pb / bw
    return {"fwd_gflop": f_fl / 1e9, "fwd_mb": f_by / 1e6, "train_gflop": (f_fl + b_fl) / 1e9,
            "train_mb": (f_by + b_by) / 1e6, "fwd_us": t_f * 1e6, "train_us": (t_f + t_b) * 1e6}


def cpu_baseline(H, W, C, P, budget_s=20.0):
    """Train-step throughput of the CPU port (oracle/unet_torch.py: torch-CPU/oneDNN restatement of the Keras
    graph, autograd, Keras-formula Adam) on this box's host cores.  Bounded sample: batch 4, >=2 timed steps."""
    import numpy as np
    import torch
    from oracle import unet_numpy as on
    from oracle import unet_torch as ot
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    ncpu = max(1, min(ncpu, int(os.environ.get("OCT_CPU_THREADS", "16"))))   # GPU-box share for one GPU is 16
    torch.set_num_threads(ncpu)
    B = 4
    cfg = on.UNetConfig(num_classes=C, pool_layers=P)
    params, state = on.init_params(cfg, seed=0, dtype=np.float32)
    tp, ts = ot.to_torch(params, state, dtype=torch.float32, requires_grad=True)
    images, labels = on.synth_scans(B, H, W, C, seed=1)
    x = torch.tensor(on.preprocess_u8(images, np.float32))
    y = torch.nn.functional.one_hot(torch.tensor(labels[..., 0].astype("int64")), C).float()
    mask = (torch.rand(B, H >> P, W >> P, 8 << P) > 0.5).float()
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
            "sample": f"{n} train steps of batch {B} ({H}x{W}x1, C={C}, pool_layers={P}, fp32) with oracle/unet_torch.py "
                      f"(torch-CPU oneDNN port of the Keras graph; TensorFlow 2.9 is not installable here)",
            "host_cpus": os.cpu_count(), "inference_ms_per_scan": round(inf_ms, 2)}


def fit_throughput(eng, H, W, C, P, B, n_scans, resident_scans_per_s):
    """The REAL training loop (reference training/training.py:358-407, common/data_generator.py:285-368): ``Model.fit``
    over a uint8 array set through ``DataGenerator`` (aug "none", shuffled, batch B): per step a host gather of B scans
    and labels, a copy into pinned buffers (three slots), H2D on a copy stream, then the same step the headline number times."""
    import numpy as np
    import torch
    from oct_image_segmentation_models_amd import optimizers
    from oct_image_segmentation_models_amd.common import custom_losses, custom_metrics
    from oct_image_segmentation_models_amd.common.data_generator import DataGenerator
    from oct_image_segmentation_models_amd.common.synthetic import make_scans
    from oct_image_segmentation_models_amd.models import get_model_class
    base_i, base_l = make_scans(64, H, W, C, seed=77)
    reps = (n_scans + 63) // 64
    images = np.tile(base_i, (reps, 1, 1, 1))[:n_scans]; labels = np.tile(base_l, (reps, 1, 1, 1))[:n_scans]
    mc = get_model_class("unet")(input_channels=1, num_classes=C, image_height=H, image_width=W, pool_layers=P)
    model = mc.build_model()
    model._device = str(eng.device)
    loss_fn = custom_losses.custom_loss_objects["dice_loss_macro"]["function"](num_classes=C, is_y_true_sparse=False)
    metric_fn = custom_metrics.training_monitor_metric_objects["dice_coef_macro"](False, C)
    model.compile(optimizer=optimizers.Adam(learning_rate=1e-3), loss=loss_fn, metrics=[metric_fn])
    gen = DataGenerator(images, labels, B, [], "none", (), False, mc.get_preprocess_input_fn(), seed=5)
    warm = DataGenerator(images[:4 * B], labels[:4 * B], B, [], "none", (), False, mc.get_preprocess_input_fn(), seed=5)
    model.fit(x=warm, epochs=1, verbose=0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    model.fit(x=gen, epochs=1, verbose=0)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n = len(gen) * B
    # the host side alone: the generator + pinned staging without any GPU work
    t1 = time.perf_counter()
    for _ in range(len(gen)):
        X, l = gen.next_batch_u8()
        model._staged(("x", 0), X); model._staged(("l", 0), l)
    host = time.perf_counter() - t1
    v = n / dt
    return {"value": round(v, 1), "unit": "B-scans/s", "scans": n, "batch": B, "epochs_timed": 1,
            "vs_resident_inputs": round(v / resident_scans_per_s, 4),
            "host_stage_scans_per_s": round(n / host, 1),
            "what": "Model.fit(DataGenerator(uint8 images, labels, aug 'none', shuffle)) -- host gather, three pinned / device "
                    "slots, H2D on a copy stream, fwd + Dice + bwd + Adam per batch; `host_stage_scans_per_s` is the generator + pinned "
                    "staging alone (one Python thread)"}


def spawn_ranks(n):
    """``bench.py --gpus N`` started as a plain script: run the N ranks as a child (no GPU call has happened yet)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def latest_pmc_file(cfg_label):
    """The committed rocprofv3 --pmc summary of the newest round for this BASELINE configuration (None otherwise)."""
    pat = {"configs[1]": "r[0-9][0-9]_pmc_traffic.json", "configs[2]": "r[0-9][0-9]_cfgC_pmc_traffic.json"}.get(cfg_label)
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pat))) if pat else []
    return files[-1] if files else None


def family(kernel_name):
    return kernel_name.split("<")[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=20,
                    help="untimed steps before the timed region (the GPU clock governor needs tens of ms from idle)")
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch (weak scaling)")
    ap.add_argument("--height", type=int, default=256)
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--classes", type=int, default=3)
    ap.add_argument("--infer-batch", type=int, default=128)
    ap.add_argument("--act-dtype", choices=["f32", "bf16"], default="f32",
                    help="f32 = the headline mode; bf16 = BASELINE configs[2] (bf16 activations and MFMA operands, "
                         "fp32 accumulation, BN statistics, parameters and optimizer)")
    ap.add_argument("--pool-layers", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-inference", action="store_true", help="train steps only (PMC passes)")
    ap.add_argument("--no-overlap", action="store_true", help="one all-reduce after the whole backward (no side stream)")
    ap.add_argument("--dump-profile", default=None, help="write the per-(kernel, layer) launch table to this JSON file")
    ap.add_argument("--force-collective", action="store_true",
                    help="N = 1: create a one-rank RCCL communicator and run the gradient all-reduces inside the timed steps")
    ap.add_argument("--no-collective-leg", action="store_true", help="N = 1: skip the extra one-rank RCCL leg")
    ap.add_argument("--no-fit", action="store_true", help="skip the Model.fit / DataGenerator throughput leg")
    ap.add_argument("--fit-scans", type=int, default=8192,
                    help="uint8 scans in the Model.fit leg's array set (one timed epoch; an epoch carries ~20 ms of fixed cost)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:      # (the rehearsal flag travels to the ranks in the environment)
        sys.exit(spawn_ranks(args.gpus))
    # stdout carries exactly ONE line, the JSON record: native libraries write banners straight to file descriptor 1 (RCCL
    # prints its version block when a communicator is created), so fd 1 points at stderr for the whole run and the record
    # goes to a private duplicate of the real stdout
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import numpy as np
    import torch
    from oct_image_segmentation_models_amd import parallel
    from oct_image_segmentation_models_amd.common.synthetic import make_scans
    from oct_image_segmentation_models_amd.engine import UNetEngine

    # OCT_BENCH_REHEARSAL=1: multi-process rehearsal on a ONE-GPU box (all ranks share cuda:0, gloo collective);
    # never used by the driver -- real runs are one rank per GPU over RCCL
    rehearsal = os.environ.get("OCT_BENCH_REHEARSAL") == "1"
    rank, local_rank, world = parallel.init("gloo" if rehearsal else "nccl", force=args.force_collective)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dev = torch.device("cuda", 0 if rehearsal else local_rank)
    torch.cuda.set_device(dev)
    B, H, W, C, P = args.batch, args.height, args.width, args.classes, args.pool_layers

    eng = UNetEngine(device=dev, input_channels=1, num_classes=C, image_height=H, image_width=W,
                     max_batch=max(B, 1 if args.no_inference else args.infer_batch), training=True,
                     seed=1000 + rank, init_seed=0, pool_layers=P,
                     dtype={"f32": "float32", "bf16": "bfloat16"}[args.act_dtype])
    parallel.broadcast_parameters(eng.params, eng.state)
    # a few distinct synthetic scans per rank, tiled to the batch (host generation is not part of the step)
    nd = min(B, 8)
    images, labels = make_scans(nd, H, W, C, seed=1234 + rank)
    reps = (B + nd - 1) // nd
    x = torch.from_numpy(np.tile(images, (reps, 1, 1, 1))[:B]).to(dev)
    lab = torch.from_numpy(np.tile(labels[..., 0], (reps, 1, 1))[:B].copy()).to(dev)
    overlap = parallel.collective_active() and not args.no_overlap
    reducer = parallel.GradReducer(eng, overlap=overlap)
    rccl = parallel.describe() if parallel.collective_active() else None

    def train_step():
        eng.forward(x, training=True, labels=lab, want_probs=False)
        loss4 = eng.loss_dice()
        reducer.backward_and_reduce(lab, macro=True, loss_scale=1.0 / world)   # bwd + (overlapped) all-reduce
        eng.adam_step(lr=1e-3)
        return loss4

    for _ in range(args.warmup):
        train_step()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    parallel.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        ev[i][0].record()
        loss4 = train_step()
        ev[i][1].record()
    torch.cuda.synchronize(); parallel.barrier()
    dt = parallel.max_over_ranks(time.perf_counter() - t0, dev)
    step_ms = sorted(a.elapsed_time(b) for a, b in ev)
    scans_per_s = world * B * args.steps / dt
    # which BASELINE.json configuration this run is: [1] = the headline (256x512x1, P=4, fp32, batch 32);
    # [2] = 512x1024x1, P=5, bf16, batch 64; anything else is labelled custom
    if (H, W, C, P, args.act_dtype, B) == (256, 512, 3, 4, "f32", 32):
        cfg_label = "configs[1]" if world == 1 else "configs[3]"
    elif (H, W, P, args.act_dtype, B) == (512, 1024, 5, "bf16", 64):
        cfg_label = "configs[2]"
    else:
        cfg_label = "custom (not a BASELINE configuration)"
    final_loss = float(loss4[0])
    mm = eng.mfma_mode_name()

    out = {
        "metric": f"B-scans/sec (train step), {H}x{W} {C}-class U-Net", "value": round(scans_per_s, 2),
        "unit": "B-scans/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32" if args.act_dtype == "f32" else "bf16 (activations + MFMA operands; f32 accumulate / BN / params)",
        "arithmetic": mm,
        "data": "synthetic",
        "config": {"workload": f"{cfg_label}: train step (fwd+Dice-macro+bwd+allreduce+Adam), per-GPU batch {B}, "
                               f"{H}x{W}x1, {C}-class, pool_layers={P}, start_neurons=8, random-init weights",
                   "global_batch": B * world, "parallelism": f"dp{world}",
                   "allreduce": ("none" if not parallel.collective_active() else
                                 ("2 segments, decoder half on a side stream under the encoder backward" if overlap
                                  else "1 flat all-reduce after backward"))},
        "step_ms_median_events": round(step_ms[len(step_ms) // 2], 4),
        "step_ms_min_events": round(step_ms[0], 4), "final_loss": round(final_loss, 5),
    }

    # ---- the exchange step through RCCL: what the library reports, and what the two all-reduces cost per step ----
    def time_collectives(red, nsteps):
        """median us per step of [all-reduce of the tail segment on the side stream] and [of the encoder segment]"""
        evs = []
        for _ in range(nsteps):
            eng.forward(x, training=True, labels=lab, want_probs=False); eng.loss_dice()
            evs.append(red.backward_and_reduce(lab, macro=True, loss_scale=1.0 / world, timed=True))
            eng.adam_step(lr=1e-3)
        torch.cuda.synchronize()
        cols = list(zip(*[[a.elapsed_time(b) * 1e3 for a, b in e] for e in evs if e]))
        return [round(sorted(c)[len(c) // 2], 2) for c in cols]

    if rccl is not None:
        out["allreduce"] = dict(rccl, in_timed_region=True, overlap=bool(overlap), grad_floats=int(eng.grads.numel()),
                                tail_offset=int(eng.grad_tail_offset()))
        out["allreduce"]["us_per_step"] = time_collectives(reducer, 10)
    elif world == 1 and not args.no_collective_leg and not rehearsal:
        try:
            import socket as _s
            with _s.socket() as sk:
                sk.bind(("127.0.0.1", 0)); os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            parallel.init("nccl", force=True)
            red1 = parallel.GradReducer(eng, overlap=True)
            for _ in range(3):
                time_collectives(red1, 1)
            torch.cuda.synchronize(); tc = time.perf_counter()
            ncol = 20
            for _ in range(ncol):
                eng.forward(x, training=True, labels=lab, want_probs=False); eng.loss_dice()
                red1.backward_and_reduce(lab, macro=True, loss_scale=1.0); eng.adam_step(lr=1e-3)
            torch.cuda.synchronize()
            ms_with = (time.perf_counter() - tc) / ncol * 1e3
            out["allreduce"] = dict(parallel.describe(), in_timed_region=False, overlap=True,
                                    grad_floats=int(eng.grads.numel()), tail_offset=int(eng.grad_tail_offset()),
                                    us_per_step=time_collectives(red1, 10), ms_per_step_with_collectives=round(ms_with, 4),
                                    note="one-rank RCCL communicator on this GPU: tail event -> all-reduce(grads[off:]) on the "
                                         "side stream -> all-reduce(grads[:off]) -> join, after the timed (collective-free) steps")
            red1.close(); parallel.shutdown()
        except Exception as e:      # noqa: BLE001  -- the leg must never cost the headline number
            out["allreduce"] = {"error": f"{type(e).__name__}: {e}"}
            try:
                parallel.shutdown()
            except Exception:      # noqa: BLE001
                pass

    # ---- inference: hipGraph-replayed forward (+argmax), batch 128, inputs resident ----
    if not args.no_inference:
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
        out["inference_ms_per_scan"] = round(infer_ms, 5)
        out["inference_batch"] = IB

    es = 4 if args.act_dtype == "f32" else 2
    cm = cost_model(H, W, C, 8, P, 2, es, PEAK_F32_TFLOPS if es == 4 else PEAK_BF16_TFLOPS)
    achieved_us = dt / args.steps / B * 1e6
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
        by_k, by_f = {}, {}
        for e in ents:
            for table, key in ((by_k, e["kernel"]), (by_f, family(e["kernel"]))):
                k = table.setdefault(key, dict(ms=0.0, flops=0.0, bytes=0.0, launches=0))
                k["ms"] += e["total_ms"]; k["flops"] += e["flops"]; k["bytes"] += e["bytes"]; k["launches"] += e["launches"]
        total_ms = sum(k["ms"] for k in by_f.values())
        # the MFMA peak a conv kernel is priced against: the f32 pipe, or -- when the engine runs its convolutions as
        # split-bf16 products on the bf16 pipe (UNetEngine.mfma_mode_name) -- the dense bf16 peak divided by the
        # number of bf16 products that make one fp32-accurate product
        prod = eng.mfma_products()
        mfma_peak = PEAK_F32_TFLOPS if prod == 0 else PEAK_BF16_TFLOPS / prod

        def roof_of(k):
            """achieved = ALGORITHMIC flops (or bytes) of the kernel's launches / their summed HIP-event duration"""
            ai = k["flops"] / max(k["bytes"], 1.0)
            if ai > mfma_peak * 1e12 / (PEAK_HBM_GBS * 1e9):
                ach = k["flops"] / (k["ms"] * 1e-3) / 1e12
                return {"bound": "mfma", "achieved": round(ach, 3), "peak": round(mfma_peak, 1), "unit": "TFLOP/s",
                        "frac": round(ach / mfma_peak, 4)}
            ach = k["bytes"] / (k["ms"] * 1e-3) / 1e9
            return {"bound": "hbm", "achieved": round(ach, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": round(ach / PEAK_HBM_GBS, 4)}

        fam, dom = max(by_f.items(), key=lambda kv: kv[1]["ms"])
        roof = roof_of(dom)
        inst = {k: v for k, v in by_k.items() if family(k) == fam}
        worst_name, worst = min(inst.items(), key=lambda kv: roof_of(kv[1])["frac"])
        # HBM traffic from the committed rocprofv3 --pmc passes of this same command (never measured by this run)
        traffic = traffic_ratio = step_traffic = None
        pmc_file = latest_pmc_file(cfg_label)
        src = None
        if pmc_file:
            try:
                pmc = json.load(open(pmc_file))
                src = f"{os.path.relpath(pmc_file, ROOT)} (kernels as of commit {pmc.get('commit', 'unknown')}; " \
                      f"committed profile, not measured by this run)"
                fams = pmc.get("families", {})
                if fam in fams:
                    traffic = round(fams[fam]["hbm_bytes_per_launch"], 0)
                if "train_step_hbm_bytes" in pmc:
                    step_traffic = pmc["train_step_hbm_bytes"]
                    traffic_ratio = round(step_traffic / (cm["train_mb"] * 1e6 * B), 3)
            except (OSError, KeyError, ValueError):
                pass
        roof.update({"traffic": traffic, "traffic_source": src, "kernel": fam, "instantiations": len(inst),
                     "measured": "HIP events recorded on the launch stream around every launch of 3 untimed steps; while "
                                 "the profiler is on every launch runs ALONE (the handle's side stream is off), so "
                                 "`achieved` is the kernel by itself -- inside a step the backward-weights kernels run "
                                 "beside the backward-data chain and both take longer (the rocprofv3 summary under "
                                 "profiles/ shows those in-step durations)",
                     "avg_launch_us": round(dom["ms"] / dom["launches"] * 1e3, 2),
                     "launches_per_step": dom["launches"] // nprof,
                     "share_of_step_kernel_time": round(dom["ms"] / total_ms, 4),
                     "algorithmic_flops_per_launch": dom["flops"] / dom["launches"],
                     "algorithmic_bytes_per_launch": dom["bytes"] / dom["launches"],
                     "worst_instantiation": dict(roof_of(worst), kernel=worst_name,
                                                 ms_per_step=round(worst["ms"] / nprof, 4)),
                     "step_frac": round(cm["train_us"] / achieved_us, 4),
                     "traffic_ratio": traffic_ratio, "step_traffic_bytes": step_traffic})
        out["roofline"] = roof
        top = sorted(by_f.items(), key=lambda kv: -kv[1]["ms"])[:12]
        out["kernel_time_ms_per_step"] = {k: round(v["ms"] / nprof, 4) for k, v in top}
        out["roofline_top_kernels"] = {k: dict(roof_of(v), ms_per_step=round(v["ms"] / nprof, 4)) for k, v in top}
    # whole-step roofline context from THIS run's shape and dtype (SURVEY A.3 formulas)
    # ... priced twice, each with ONE set of peaks: (a) SURVEY 8(d)'s model (the arithmetic the reference's dtype implies:
    # f32 MFMA 157.3 TF or bf16 2500 TF; HBM 6.29 TB/s measured by a copy); (b) the arithmetic this engine actually runs
    # (f32 mode: 6 bf16 products per product -> 2500 / 6 = 416.7 TF) against the HBM SPEC rate of 8 TB/s -- the peaks
    # `roofline` / `roofline_top_kernels` use for single kernels
    prod_used = eng.mfma_products()
    used_peak = PEAK_F32_TFLOPS if prod_used == 0 else PEAK_BF16_TFLOPS / prod_used
    cm_used = cost_model(H, W, C, 8, P, 2, es, used_peak, PEAK_HBM_GBS)
    out["step_vs_roofline"] = {"algorithmic_gflop_per_scan": round(cm["train_gflop"], 3),
                               "algorithmic_mb_per_scan": round(cm["train_mb"], 1),
                               "achieved_us_per_scan": round(achieved_us, 2),
                               "survey_model": {"peaks": f"{'f32 MFMA 157.3' if es == 4 else 'bf16 MFMA 2500'} TFLOP/s, HBM 6.29 TB/s (measured copy rate)",
                                                "roofline_us_per_scan": round(cm["train_us"], 2),
                                                "frac": round(cm["train_us"] / achieved_us, 4),
                                                "inference_roofline_us_per_scan": round(cm["fwd_us"], 2)},
                               "arithmetic_used": {"peaks": f"MFMA {used_peak:.1f} TFLOP/s ({mm.split(':')[0]}), HBM 8.0 TB/s (spec)",
                                                   "roofline_us_per_scan": round(cm_used["train_us"], 2),
                                                   "frac": round(cm_used["train_us"] / achieved_us, 4),
                                                   "inference_roofline_us_per_scan": round(cm_used["fwd_us"], 2)},
                               # (kept for readers of earlier rounds' records: the SURVEY-model figures)
                               "roofline_us_per_scan": round(cm["train_us"], 2),
                               "frac": round(cm["train_us"] / achieved_us, 4),
                               "inference_roofline_us_per_scan": round(cm["fwd_us"], 2)}
    if rank == 0 and world == 1 and not args.no_inference and os.environ.get("OCT_BENCH_NO_E2E") != "1":
        try:
            from oct_image_segmentation_models_amd.evaluation import pipeline
            out.update(pipeline.bench_fields(eng, images, C, labels_u8=labels[..., 0]))
        except ImportError:
            pass
    if rank == 0 and world == 1 and not args.no_fit and not args.no_inference and args.act_dtype == "f32":
        try:
            out["fit"] = fit_throughput(eng, H, W, C, P, B, args.fit_scans, scans_per_s)
        except Exception as e:      # noqa: BLE001
            out["fit"] = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(H, W, C, P)
    if rank == 0:
        if rehearsal:
            out["data"] = "synthetic (REHEARSAL: ranks share one GPU, gloo collective -- not a measurement)"
        real_stdout.write(json.dumps(out) + "\n"); real_stdout.flush()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
