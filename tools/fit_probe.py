#!/usr/bin/env python3
"""Runs on the GPU box: where one Model.fit step spends its HOST time (the GPU step is ~3.9 ms at cfg-A, B = 32).
usage: tools/fit_probe.py [--pre-train] [--graph] [--profile]   (phases bench.py runs on its own engine before the fit leg)"""
import sys, time, collections
import numpy as np
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from oct_image_segmentation_models_amd import optimizers
from oct_image_segmentation_models_amd.common import custom_losses, custom_metrics
from oct_image_segmentation_models_amd.common.data_generator import DataGenerator
from oct_image_segmentation_models_amd.common.synthetic import make_scans
from oct_image_segmentation_models_amd.models import get_model_class
from oct_image_segmentation_models_amd.models import engine_model

H, W, C, P, B, N = 256, 512, 3, 4, 32, int(__import__('os').environ.get('FIT_SCANS', '2048'))
base_i, base_l = make_scans(64, H, W, C, seed=77)
images = np.tile(base_i, (N // 64, 1, 1, 1)); labels = np.tile(base_l, (N // 64, 1, 1, 1))
mc = get_model_class("unet")(input_channels=1, num_classes=C, image_height=H, image_width=W, pool_layers=P)
model = mc.build_model()
loss_fn = custom_losses.custom_loss_objects["dice_loss_macro"]["function"](num_classes=C, is_y_true_sparse=False)
metric_fn = custom_metrics.training_monitor_metric_objects["dice_coef_macro"](False, C)
model.compile(optimizer=optimizers.Adam(learning_rate=1e-3), loss=loss_fn, metrics=[metric_fn])
gen = DataGenerator(images, labels, B, [], "none", (), False, mc.get_preprocess_input_fn(), seed=5)
warm = DataGenerator(images[:4 * B], labels[:4 * B], B, [], "none", (), False, mc.get_preprocess_input_fn(), seed=5)
model.fit(x=warm, epochs=1, verbose=0)
torch.cuda.synchronize()
# optional: what bench.py does with ITS engine before the fit leg (another handle in the same process)
if any(a in sys.argv for a in ("--pre-train", "--graph", "--profile")):
    from oct_image_segmentation_models_amd.engine import UNetEngine
    dev = model._engine.device
    engA = UNetEngine(device=dev, input_channels=1, num_classes=C, image_height=H, image_width=W, max_batch=128, training=True,
                      seed=1000, init_seed=0, pool_layers=P, dtype="float32")
    xa = torch.from_numpy(images[:B]).to(dev); la = torch.from_numpy(labels[:B, ..., 0].copy()).to(dev)
    def stepA():
        engA.forward(xa, training=True, labels=la, want_probs=False); engA.loss_dice(); engA.backward(la, macro=True, loss_scale=1.0); engA.adam_step(lr=1e-3)
    if "--pre-train" in sys.argv:
        for _ in range(25): stepA()
    if "--graph" in sys.argv:
        xi = torch.from_numpy(np.tile(images[:32], (4, 1, 1, 1))).to(dev)
        engA.graph_capture(xi, want_probs=True, want_argmax=True)
        for _ in range(13): engA.graph_launch()
    if "--profile" in sys.argv:
        engA.profile_begin()
        for _ in range(3):
            engA.forward(xa, training=True, labels=la, want_probs=False); engA.loss_dice(); engA.backward(la, macro=True, loss_scale=1.0)
        engA.profile_end()
    torch.cuda.synchronize()

acc = collections.defaultdict(float)
def timed(obj, name, key):
    f = getattr(obj, name)
    def w(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); acc[key] += time.perf_counter() - t; return r
    setattr(obj, name, w)
timed(model, "_host_batch", "host_batch"); timed(model, "_upload", "upload"); timed(model, "_release", "release")
eng = model._engine
timed(eng, "forward", "forward"); timed(eng, "loss_dice", "loss")
timed(model.optimizer, "apply", "adam")
import oct_image_segmentation_models_amd.parallel as par
orig = par.GradReducer.backward_and_reduce
def bw(self, *a, **k):
    t = time.perf_counter(); r = orig(self, *a, **k); acc["backward"] += time.perf_counter() - t; return r
par.GradReducer.backward_and_reduce = bw
step_ev = []
_apply = model.optimizer.apply
def apply_and_mark(e):
    r = _apply(e); ev = torch.cuda.Event(enable_timing=True); ev.record(); step_ev.append(ev); return r
model.optimizer.apply = apply_and_mark
t0 = time.perf_counter()
model.fit(x=gen, epochs=1, verbose=0)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
n = len(gen)
print(f"{n} steps, {dt / n * 1e3:.3f} ms per step, {n * B / dt:.0f} scans/s")
d = sorted(a.elapsed_time(b) for a, b in zip(step_ev, step_ev[1:]))
print(f"   GPU time between the ends of consecutive steps: median {d[len(d) // 2]:.3f} ms, min {d[0]:.3f}, max {d[-1]:.3f}, sum {sum(d):.1f} ms of {dt * 1e3:.1f} ms wall; largest {[round(x, 2) for x in d[-6:]]}")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"   host time in {k:12s} {v / n * 1e3:7.3f} ms per step")
