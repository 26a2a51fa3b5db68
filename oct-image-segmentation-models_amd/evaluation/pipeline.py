"""The inference-only path of BASELINE configs[4]: ``evaluate_model`` / ``predict`` at device batch 128 --
a hipGraph-captured forward over fixed buffers, raw uint8 images uploaded from pinned double buffers on a copy
stream, uint8 arg-max class maps and uint8 boundary maps (not fp32 probabilities) downloaded into pinned double
buffers, the host min-path post-process fanned out to a process pool; under ``torchrun`` the test set is sharded by
``parallel.shard_range`` (no collective).  Reference: evaluation/evaluation.py:108-135,289-315 and
prediction/prediction.py:70-81,134-143 (one ``predict`` call and one graph build per image there).

Results are bit-identical to the per-image path (``Model.predict_labels``): inference is independent of batch
composition (tests/test_gpu_workflow.py::test_batched_pipeline_equals_per_image_path)."""
from __future__ import annotations

import time
from typing import Iterator, Optional, Tuple

import numpy as np
import torch

from ..min_path_processing.pool import SegmentPool, default_workers


class BatchedPredictor:
    """Fixed-batch graph replay with overlapped transfers.  ``run(images_u8)`` yields
    ``(lo, hi, labels (n,H,W) uint8, maps (n,C-1,H,W) uint8 | None)`` per device batch, in order."""

    def __init__(self, engine, batch: int, want_maps: bool = True, bg_ilm: bool = True, bg_csi: bool = False):
        if not 1 <= batch <= engine.cfg.max_batch:
            raise ValueError(f"batch {batch} outside 1..max_batch={engine.cfg.max_batch}")
        self.eng, self.B, self.want_maps, self.bg = engine, int(batch), want_maps, (bg_ilm, bg_csi)
        dev, H, W, C = engine.device, engine.cfg.H, engine.cfg.W, engine.cfg.n_cls
        ic = engine.cfg.in_ch
        self.x_dev = torch.zeros((self.B, H, W, ic), dtype=torch.uint8, device=dev)       # the graph's fixed input
        self.x_stage = [torch.empty_like(self.x_dev) for _ in range(2)]                    # H2D landing buffers
        self.x_pin = [torch.empty((self.B, H, W, ic), dtype=torch.uint8).pin_memory() for _ in range(2)]
        self.lab_pin = [torch.empty((self.B, H, W), dtype=torch.uint8).pin_memory() for _ in range(2)]
        self.map_pin = [torch.empty((self.B, C - 1, H, W), dtype=torch.uint8).pin_memory() for _ in range(2)] if want_maps else None
        self.lab_dev = [torch.empty((self.B, H, W), dtype=torch.uint8, device=dev) for _ in range(2)]
        self.map_dev = [torch.empty((self.B, C - 1, H, W), dtype=torch.uint8, device=dev) for _ in range(2)] if want_maps else None
        self.copy_in = torch.cuda.Stream(device=dev)
        self.copy_out = torch.cuda.Stream(device=dev)
        _, self.am = engine.graph_capture(self.x_dev, want_probs=False, want_argmax=True)

    def run(self, images_u8: np.ndarray) -> Iterator[Tuple[int, int, np.ndarray, Optional[np.ndarray]]]:
        images_u8 = np.ascontiguousarray(images_u8)
        if images_u8.dtype != np.uint8:
            raise TypeError("the batched pipeline takes raw uint8 images (the /255 happens on the device)")
        n, B, eng = images_u8.shape[0], self.B, self.eng
        main = torch.cuda.current_stream(eng.device)
        nb = (n + B - 1) // B
        up_done = [torch.cuda.Event() for _ in range(2)]
        x_free = [torch.cuda.Event() for _ in range(2)]
        out_done = [torch.cuda.Event() for _ in range(2)]
        out_ready = [torch.cuda.Event() for _ in range(2)]

        def upload(i):
            lo, hi, s = i * B, min(n, (i + 1) * B), i & 1
            if i >= 2:
                # the async H2D of batch i-2 READS this pinned buffer: the host must not overwrite it before that copy has
                # run (the device-side wait below only protects the staging buffer; no other host sync orders copy_in)
                up_done[s].synchronize()
            self.x_pin[s][:hi - lo].copy_(torch.from_numpy(images_u8[lo:hi]))       # host gather into pinned memory
            with torch.cuda.stream(self.copy_in):
                if i >= 2:
                    self.copy_in.wait_event(x_free[s])                                  # staging buffer consumed by batch i-2
                self.x_stage[s][:hi - lo].copy_(self.x_pin[s][:hi - lo], non_blocking=True)
                up_done[s].record(self.copy_in)

        pending = None
        if nb:
            upload(0)
        for i in range(nb):
            lo, hi, s = i * B, min(n, (i + 1) * B), i & 1
            if i + 1 < nb:
                upload(i + 1)                                                           # overlaps the forward of batch i
            main.wait_event(up_done[s])
            self.x_dev.copy_(self.x_stage[s])                                           # D2D, then the staging buffer is free
            x_free[s].record(main)
            eng.graph_launch()
            if i >= 2:
                main.wait_event(out_done[s])                                            # device out buffers of batch i-2 downloaded
            self.lab_dev[s].copy_(self.am)
            if self.want_maps:
                self.map_dev[s].copy_(eng.boundary_maps(self.am, bg_ilm=self.bg[0], bg_csi=self.bg[1]))
            out_ready[s].record(main)
            with torch.cuda.stream(self.copy_out):
                self.copy_out.wait_event(out_ready[s])
                self.lab_pin[s].copy_(self.lab_dev[s], non_blocking=True)
                if self.want_maps:
                    self.map_pin[s].copy_(self.map_dev[s], non_blocking=True)
                out_done[s].record(self.copy_out)
            if pending is not None:
                yield self._collect(*pending)
            pending = (lo, hi, s, out_done[s])
        if pending is not None:
            yield self._collect(*pending)

    def _collect(self, lo, hi, s, ev):
        ev.synchronize()
        labels = self.lab_pin[s][:hi - lo].numpy().copy()
        maps = self.map_pin[s][:hi - lo].numpy().copy() if self.want_maps else None
        return lo, hi, labels, maps


def bench_fields(engine, images_u8: np.ndarray, num_classes: int, batch: Optional[int] = None, n_batches: int = 6,
                 labels_u8: Optional[np.ndarray] = None) -> dict:
    """``bench.py`` fields for BASELINE configs[4] / BASELINE.md 5.4: host post-process cost (1 thread and the pool) and
    end-to-end inference ms per B-scan (upload + graph forward + boundary maps + download + pooled min-path), beside the
    GPU-only figure the bench already reports.

    What the graph search costs depends on its input: the bench's network is randomly initialised and a few steps old, its
    boundary maps are noise, and Dijkstra over noise is 5-20x slower than over the single clean ridge a trained model
    emits (and varies from run to run with the weights).  With ``labels_u8`` (the synthetic ground-truth class maps of
    the same scans) the headline figures use the boundary maps OF THOSE LABELS -- computed on the device by the same
    ``oct_boundary_maps`` kernel -- as the post-process input, while the GPU side still runs the full pipeline on the
    images; the figures for the network's own (noise) maps are reported beside them as ``*_untrained_maps``."""
    B = int(batch or engine.cfg.max_batch)
    H, W = engine.cfg.H, engine.cfg.W
    reps = (B * n_batches + images_u8.shape[0] - 1) // images_u8.shape[0]
    imgs = np.tile(images_u8, (reps, 1, 1, 1))[:B * n_batches]
    workers = default_workers()
    pred = BatchedPredictor(engine, B, want_maps=True)
    clean = None
    if labels_u8 is not None:
        lab = np.ascontiguousarray(np.tile(labels_u8.reshape((-1, H, W)), (reps, 1, 1))[:B])
        clean = engine.boundary_maps(torch.from_numpy(lab).to(engine.device)).cpu().numpy()
    out = {}
    with SegmentPool((H, W), gsgrad=1, workers=workers) as pool, SegmentPool((H, W), gsgrad=1, workers=1) as solo:
        first = next(iter(pred.run(imgs[:B])))                      # warm-up: graph, pinned buffers, worker start-up
        maps0 = first[3]
        pool.segment(maps0[:min(B, 2 * workers)])
        ns = min(B, 8)

        def host_cost(maps):
            t0 = time.perf_counter(); solo.segment(maps[:ns]); t1 = time.perf_counter()
            pool.segment(maps); t2 = time.perf_counter()
            return round((t1 - t0) / ns * 1e3, 3), round((t2 - t1) / B * 1e3, 4)

        def e2e(maps_for_pool):
            # GPU batches pipelined against the pool (post-process of batch i runs while batch i+1 is on the GPU)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            jobs = []
            for lo, hi, labels, maps in pred.run(imgs):
                jobs.append(pool.segment_async(maps if maps_for_pool is None else maps_for_pool[:hi - lo]))
            for j in jobs:
                j.get()
            return round((time.perf_counter() - t0) / imgs.shape[0] * 1e3, 4)

        one, pooled = host_cost(maps0 if clean is None else clean)
        out["host_postprocess"] = {"what": f"segment_maps over the {num_classes - 1} boundary maps of one {H}x{W} B-scan "
                                           "(native liboct_minpath.so Dijkstra, identical results to the reference)",
                                   "maps": "the network's own" if clean is None else "boundary maps of the synthetic ground-truth class maps (what a trained model emits)",
                                   "ms_per_scan_1_thread": one, "pool_workers": workers, "ms_per_scan_pool": pooled}
        out["inference_e2e_ms_per_scan"] = e2e(clean)
        if clean is not None:
            one_u, pooled_u = host_cost(maps0)
            out["host_postprocess"]["untrained_maps"] = {"ms_per_scan_1_thread": one_u, "ms_per_scan_pool": pooled_u}
            out["inference_e2e_ms_per_scan_untrained_maps"] = e2e(None)
        t0 = time.perf_counter()
        for _ in pred.run(imgs):
            pass
        dt = time.perf_counter() - t0
        out["inference_gpu_pipeline_ms_per_scan"] = round(dt / imgs.shape[0] * 1e3, 4)
        out["inference_e2e"] = {"batch": B, "scans": int(imgs.shape[0]), "stages": "pinned u8 upload -> hipGraph forward + "
                                "arg-max -> boundary maps -> u8 download -> pooled segment_maps (BASELINE configs[4])"}
    return out
