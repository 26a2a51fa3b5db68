"""``UNetEngine``: Python owner of one ``oct_unet`` handle (one per GPU rank).

PyTorch-ROCm is used for plumbing only -- device allocations, the current HIP
stream, and (in ``parallel.py``) ``torch.distributed``/RCCL on the flat gradient
tensor.  All arithmetic of the hot path happens inside ``liboct_unet_hip.so``.

Replaces, for the ``"unet"`` architecture, what the reference obtains from
``tf.keras.Model`` (reference call sites: training/training.py:243-266,401-407;
evaluation/evaluation.py:129-135; prediction/prediction.py:75-81).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _hip
from ._hip import OctError, UNetCfg, UNetIO, LayerInfo


def _require_gpu(device) -> torch.device:
    dev = torch.device(device)
    if dev.type != "cuda" or not torch.cuda.is_available():
        raise OctError("the OCT U-Net engine needs an AMD GPU (torch.cuda.is_available() is False); "
                       "there is no CPU fallback")
    return dev


def make_cfg(*, input_channels: int, num_classes: int, image_height: int, image_width: int,
             start_neurons: int = 8, pool_layers: int = 4, conv_layers: int = 2,
             enc_kernel: Sequence[int] = (3, 3), dec_kernel: Sequence[int] = (2, 2),
             max_batch: int = 1, training: bool = False, seed: int = 0,
             bn_eps: float = 1e-3, bn_momentum: float = 0.99, dropout_rate: float = 0.5,
             bn_unbiased_moving_var: bool = True, dtype="float32") -> UNetCfg:
    ek, dk = tuple(enc_kernel), tuple(dec_kernel)
    if ek[0] != ek[1] or dk[0] != dk[1]:
        raise OctError("only square kernels are supported")
    cfg = UNetCfg()
    _hip.lib().oct_unet_cfg_default(C.byref(cfg))
    cfg.in_ch, cfg.n_cls, cfg.H, cfg.W = input_channels, num_classes, image_height, image_width
    cfg.max_batch, cfg.start_neurons, cfg.pool_layers, cfg.conv_layers = max_batch, start_neurons, pool_layers, conv_layers
    dmap = {"float32": 0, "f32": 0, 0: 0, "bfloat16": 1, "bf16": 1, 1: 1}
    if dtype not in dmap:
        raise OctError(f"dtype must be 'float32' or 'bfloat16' (activation storage), got {dtype!r}")
    cfg.enc_k, cfg.dec_k, cfg.dtype, cfg.training = ek[0], dk[0], dmap[dtype], int(training)
    cfg.bn_eps, cfg.bn_momentum, cfg.dropout_rate = bn_eps, bn_momentum, dropout_rate
    cfg.bn_unbiased_moving_var, cfg.seed = int(bn_unbiased_moving_var), seed & 0xFFFFFFFFFFFFFFFF
    _hip.check(_hip.lib().oct_unet_cfg_check(C.byref(cfg)), "oct_unet_cfg_check")
    return cfg


def layer_table(cfg: UNetCfg) -> List[dict]:
    """Conv(+BN) nodes in Keras creation order with their flat-buffer offsets (host only, no GPU)."""
    l = _hip.lib()
    out = []
    for i in range(l.oct_unet_layer_count(C.byref(cfg))):
        info = LayerInfo()
        _hip.check(l.oct_unet_layer_info(C.byref(cfg), i, C.byref(info)), "oct_unet_layer_info")
        out.append({k: (getattr(info, k).decode() if k == "name" else int(getattr(info, k))) for k, _ in LayerInfo._fields_})
    return out


def glorot_init(cfg: UNetCfg, seed: int = 0):
    """Keras initial values (glorot_uniform kernels, zero bias, gamma=1, beta=0, moving mean 0 / var 1)."""
    rng = np.random.default_rng(seed)
    l = _hip.lib()
    params = np.zeros(l.oct_unet_param_count(C.byref(cfg)), np.float32)
    state = np.zeros(l.oct_unet_state_count(C.byref(cfg)), np.float32)
    for L in layer_table(cfg):
        n = L["kh"] * L["kw"] * L["cin"] * L["cout"]
        lim = np.sqrt(6.0 / (L["kh"] * L["kw"] * (L["cin"] + L["cout"])))
        params[L["kernel_off"]:L["kernel_off"] + n] = rng.uniform(-lim, lim, n).astype(np.float32)
        if L["has_bn"]:
            params[L["gamma_off"]:L["gamma_off"] + L["cout"]] = 1.0
            state[L["moving_var_off"]:L["moving_var_off"] + L["cout"]] = 1.0
    return params, state


class UNetEngine:
    def __init__(self, *, device="cuda:0", init_seed: int = 0, **cfg_kwargs):
        self.device = _require_gpu(device)
        self.cfg = make_cfg(**cfg_kwargs)
        self.layers = layer_table(self.cfg)
        l = _hip.lib()
        self.n_params = int(l.oct_unet_param_count(C.byref(self.cfg)))
        self.n_state = int(l.oct_unet_state_count(C.byref(self.cfg)))
        ws_bytes = int(l.oct_unet_workspace_bytes(C.byref(self.cfg)))
        with torch.cuda.device(self.device):
            p0, s0 = glorot_init(self.cfg, init_seed)
            self.params = torch.from_numpy(p0).to(self.device)
            self.state = torch.from_numpy(s0).to(self.device)
            self.grads = torch.zeros(self.n_params, dtype=torch.float32, device=self.device) if self.cfg.training else None
            self.workspace = torch.empty(ws_bytes, dtype=torch.uint8, device=self.device)
            self._loss4 = torch.zeros(4, dtype=torch.float32, device=self.device)
            h = C.c_void_p()
            _hip.check(l.oct_unet_create(C.byref(self.cfg), self.params.data_ptr(),
                                         self.grads.data_ptr() if self.grads is not None else None,
                                         self.state.data_ptr(), self.workspace.data_ptr(), ws_bytes, C.byref(h)),
                       "oct_unet_create")
        self._h = h
        self._opt: Dict[str, torch.Tensor] = {}
        self.opt_step = 0
        self._keep = []  # tensors referenced by an in-flight / captured launch

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                _hip.lib().oct_unet_destroy(h)
            except Exception:   # interpreter teardown: module globals may already be gone
                pass
            self._h = None

    # ---- plumbing --------------------------------------------------------------------------------
    @property
    def H(self): return self.cfg.H
    @property
    def W(self): return self.cfg.W
    @property
    def num_classes(self): return self.cfg.n_cls

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _check_x(self, x: torch.Tensor):
        if x.device != self.device or not x.is_contiguous():
            raise OctError("input must be a contiguous tensor on the engine's device")
        if x.dim() != 4 or tuple(x.shape[1:]) != (self.cfg.H, self.cfg.W, self.cfg.in_ch):
            raise OctError(f"input must be (B,{self.cfg.H},{self.cfg.W},{self.cfg.in_ch}), got {tuple(x.shape)}")
        if x.dtype not in (torch.uint8, torch.float32):
            raise OctError("input must be uint8 (raw) or float32 (already /255)")
        if not 1 <= x.shape[0] <= self.cfg.max_batch:
            raise OctError(f"batch {x.shape[0]} outside 1..max_batch={self.cfg.max_batch}")

    def _check_labels(self, labels: torch.Tensor, B: int):
        if labels.device != self.device or labels.dtype != torch.uint8 or not labels.is_contiguous() \
                or labels.numel() != B * self.cfg.H * self.cfg.W:
            raise OctError("labels must be a contiguous uint8 (B,H,W[,1]) tensor on the engine's device")

    def _io(self, B, labels, want_probs, want_argmax, probs_out=None, argmax_out=None):
        probs = am = None
        if want_probs:
            probs = probs_out if probs_out is not None else torch.empty(
                (B, self.cfg.H, self.cfg.W, self.cfg.n_cls), dtype=torch.float32, device=self.device)
        if want_argmax:
            am = argmax_out if argmax_out is not None else torch.empty(
                (B, self.cfg.H, self.cfg.W), dtype=torch.uint8, device=self.device)
        io = UNetIO(probs.data_ptr() if probs is not None else None, am.data_ptr() if am is not None else None,
                    labels.data_ptr() if labels is not None else None)
        return io, probs, am

    # ---- hot path --------------------------------------------------------------------------------
    def forward(self, x: torch.Tensor, training: bool = False, labels: Optional[torch.Tensor] = None,
                want_probs: bool = True, want_argmax: bool = False, probs_out=None, argmax_out=None):
        self._check_x(x)
        B = x.shape[0]
        if labels is not None:
            self._check_labels(labels, B)
        io, probs, am = self._io(B, labels, want_probs, want_argmax, probs_out, argmax_out)
        with torch.cuda.device(self.device):
            _hip.check(_hip.lib().oct_unet_forward(self._h, x.data_ptr(), int(x.dtype == torch.uint8), B,
                                                   int(training), C.byref(io), self._stream()), "oct_unet_forward")
        self._keep = [x, labels, probs, am]
        return probs, am

    def loss_dice(self, smooth: float = 1e-5) -> torch.Tensor:
        """[dice_loss_macro, dice_loss_micro, dice_coef_macro, dice_coef_micro] of the last forward (device tensor)."""
        out = torch.empty(4, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _hip.check(_hip.lib().oct_unet_loss_dice(self._h, smooth, out.data_ptr(), self._stream()), "oct_unet_loss_dice")
        return out

    def set_focal_dice(self, focal_loss_weight: float = 0.5, gamma: float = 2.0, class_weight=None) -> None:
        """Select ``focal_dice_loss`` (reference custom_losses.py:98-178) for the following forward / loss / backward
        calls: L = w*focal + (1-w)*dice.  ``focal_loss_weight = 0`` restores the plain Dice losses."""
        cw = None
        if class_weight is not None:
            cw = torch.as_tensor(np.asarray(class_weight, np.float32), device=self.device).contiguous()
            if cw.numel() != self.cfg.n_cls:
                raise OctError(f"class_weight must have {self.cfg.n_cls} entries")
        self._focal_cw = cw          # keep the device buffer alive: the library only stores the pointer
        self._focal_active = float(focal_loss_weight) > 0.0
        with torch.cuda.device(self.device):
            _hip.check(_hip.lib().oct_unet_set_focal_dice(self._h, float(focal_loss_weight), float(gamma),
                                                          cw.data_ptr() if cw is not None else None), "oct_unet_set_focal_dice")

    def loss_focal_dice(self, smooth: float = 1e-5) -> torch.Tensor:
        """loss_dice() + [focal term, w*focal+(1-w)*dice_macro, w*focal+(1-w)*dice_micro, 0] (device tensor, 8 floats)."""
        out = torch.empty(8, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _hip.check(_hip.lib().oct_unet_loss_focal_dice(self._h, smooth, out.data_ptr(), self._stream()), "oct_unet_loss_focal_dice")
        return out

    def backward(self, labels: torch.Tensor, macro: bool = True, loss_scale: float = 1.0):
        self._check_labels(labels, labels.shape[0])
        with torch.cuda.device(self.device):
            _hip.check(_hip.lib().oct_unet_backward(self._h, labels.data_ptr(), int(macro), loss_scale, self._stream()),
                       "oct_unet_backward")

    def adam_step(self, lr=1e-3, beta_1=0.9, beta_2=0.999, epsilon=1e-7):
        if "m" not in self._opt:
            self._opt["m"] = torch.zeros_like(self.params); self._opt["v"] = torch.zeros_like(self.params)
        self.opt_step += 1
        with torch.cuda.device(self.device):
            _hip.check(_hip.lib().oct_adam_step(self.params.data_ptr(), self.grads.data_ptr(), self._opt["m"].data_ptr(),
                                                self._opt["v"].data_ptr(), self.n_params, lr, beta_1, beta_2, epsilon,
                                                self.opt_step, self._stream()), "oct_adam_step")

    def sgd_step(self, lr=1e-2, momentum=0.0):
        mom = None
        if momentum != 0.0:
            if "mom" not in self._opt:
                self._opt["mom"] = torch.zeros_like(self.params)
            mom = self._opt["mom"].data_ptr()
        self.opt_step += 1
        with torch.cuda.device(self.device):
            _hip.check(_hip.lib().oct_sgd_step(self.params.data_ptr(), self.grads.data_ptr(), mom, self.n_params, lr,
                                               momentum, self._stream()), "oct_sgd_step")

    # ---- data-parallel overlap hook ---------------------------------------------------
    def grad_tail_offset(self) -> int:
        """First float of the gradient segment (bottleneck + decoder + head) that is final at the tail event."""
        return int(_hip.lib().oct_unet_grad_tail_offset(C.byref(self.cfg)))

    def set_tail_event(self, event: Optional[torch.cuda.Event]) -> None:
        """``backward`` records ``event`` on its stream once grads[grad_tail_offset():] are final (None disables)."""
        if event is not None and not event.cuda_event:
            with torch.cuda.device(self.device):
                event.record()          # torch creates the HIP event lazily, at its first record
        self._tail_event = event        # keep the handle alive
        _hip.check(_hip.lib().oct_unet_set_tail_event(self._h, C.c_void_p(event.cuda_event) if event is not None else None),
                   "oct_unet_set_tail_event")

    # ---- arithmetic mode of the convolution kernels (reported by bench.py) --------------------------
    def mfma_products(self) -> int:
        """0: the convolutions run on the f32 MFMA pipe.  n > 0: on the bf16 pipe, n bf16 products per product
        (6 = exact three-term split of both fp32 operands, csrc/kernels_bx.hpp; 1 = bf16 operands, cfg.dtype 1)."""
        if not self.handle_option("mfma_mode"):
            return 0
        return 1 if self.cfg.dtype == 1 else 6

    def mfma_mode_name(self) -> str:
        n = self.mfma_products()
        if n == 0:
            return "f32 MFMA (v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32), f32 accumulate"
        if n == 1:
            return "bf16 MFMA (v_mfma_f32_32x32x16_bf16 / 16x16x32), bf16 operands, f32 accumulate"
        return ("fp32 via exact 3-term bf16 split: 6 bf16 MFMAs (v_mfma_f32_32x32x16_bf16 / 16x16x32) per fp32 product, "
                "f32 accumulate (first layer / head: f32 VALU)")

    # ---- per-launch profiler ---------------------------------------------------------------------
    def profile_begin(self):
        _hip.check(_hip.lib().oct_unet_profile_begin(self._h), "oct_unet_profile_begin")

    def profile_end(self) -> List[dict]:
        """One dict per (kernel instantiation, layer): launches, total_ms, algorithmic flops and bytes."""
        n = C.c_int(0)
        buf = (_hip.ProfileEntry * 1024)()
        with torch.cuda.device(self.device):
            _hip.check(_hip.lib().oct_unet_profile_end(self._h, buf, 1024, C.byref(n)), "oct_unet_profile_end")
        return [dict(kernel=buf[i].kernel.decode(), layer=buf[i].layer.decode(), launches=buf[i].launches,
                     total_ms=buf[i].total_ms, flops=buf[i].flops, bytes=buf[i].bytes) for i in range(min(n.value, 1024))]

    # ---- dropout replay (tests) -------------------------------------------------------------------
    def set_dropout_step(self, step: int):
        _hip.check(_hip.lib().oct_unet_set_dropout_step(self._h, step), "oct_unet_set_dropout_step")

    def dropout_mask(self, B: int) -> torch.Tensor:
        P = self.cfg.pool_layers
        m = torch.empty((B, self.cfg.H >> P, self.cfg.W >> P, self.cfg.start_neurons << P), dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            _hip.check(_hip.lib().oct_unet_dropout_mask(self._h, B, m.data_ptr(), self._stream()), "oct_unet_dropout_mask")
        return m

    # ---- inference hipGraph ------------------------------------------------------------------------
    def graph_capture(self, x: torch.Tensor, want_probs=True, want_argmax=False):
        """Capture one inference forward over fixed buffers; returns (probs, argmax) output tensors that every
        ``graph_launch`` refills.  ``x`` must be refilled in place (``x.copy_``) between launches."""
        self._check_x(x)
        io, probs, am = self._io(x.shape[0], None, want_probs, want_argmax)
        self._graph_keep = [x, probs, am]
        s = torch.cuda.Stream(self.device)
        s.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.device(self.device):
            _hip.check(_hip.lib().oct_unet_graph_capture(self._h, x.data_ptr(), int(x.dtype == torch.uint8), x.shape[0],
                                                         C.byref(io), C.c_void_p(s.cuda_stream)), "oct_unet_graph_capture")
        torch.cuda.current_stream(self.device).wait_stream(s)
        return probs, am

    def graph_launch(self):
        with torch.cuda.device(self.device):
            _hip.check(_hip.lib().oct_unet_graph_launch(self._h, self._stream()), "oct_unet_graph_launch")

    # ---- post-step on device ---------------------------------------------------------------------
    def boundary_maps(self, labels: torch.Tensor, bg_ilm: bool = True, bg_csi: bool = False) -> torch.Tensor:
        """(B,H,W) uint8 class maps (e.g. the arg-max output) -> (B, num_classes-1, H, W) uint8 boundary maps."""
        if labels.device != self.device or labels.dtype != torch.uint8 or not labels.is_contiguous() or labels.dim() != 3:
            raise OctError("labels must be a contiguous uint8 (B,H,W) tensor on the engine's device")
        B, H, W = labels.shape
        out = torch.empty((B, self.cfg.n_cls - 1, H, W), dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            _hip.check(_hip.lib().oct_boundary_maps(labels.data_ptr(), B, H, W, self.cfg.n_cls, int(bg_ilm), int(bg_csi),
                                                    out.data_ptr(), self._stream()), "oct_boundary_maps")
        return out

    # ---- weights exchange --------------------------------------------------------------------------
    def get_weights(self) -> List[np.ndarray]:
        """Keras ``get_weights()`` order: Conv2D [kernel HWIO, bias]; BN [gamma, beta, moving_mean, moving_var]."""
        p = self.params.cpu().numpy(); s = self.state.cpu().numpy()
        out = []
        for L in self.layers:
            n = L["kh"] * L["kw"] * L["cin"] * L["cout"]; c = L["cout"]
            out.append(p[L["kernel_off"]:L["kernel_off"] + n].reshape(L["kh"], L["kw"], L["cin"], c).copy())
            out.append(p[L["bias_off"]:L["bias_off"] + c].copy())
            if L["has_bn"]:
                out += [p[L["gamma_off"]:L["gamma_off"] + c].copy(), p[L["beta_off"]:L["beta_off"] + c].copy(),
                        s[L["moving_mean_off"]:L["moving_mean_off"] + c].copy(), s[L["moving_var_off"]:L["moving_var_off"] + c].copy()]
        return out

    def set_weights(self, weights: Sequence[np.ndarray]):
        p = np.empty(self.n_params, np.float32); s = np.empty(self.n_state, np.float32)
        it = iter(weights)
        try:
            for L in self.layers:
                n = L["kh"] * L["kw"] * L["cin"] * L["cout"]; c = L["cout"]
                k = np.asarray(next(it), np.float32)
                if k.shape != (L["kh"], L["kw"], L["cin"], c):
                    raise OctError(f"{L['name']}: kernel shape {k.shape} != {(L['kh'], L['kw'], L['cin'], c)}")
                p[L["kernel_off"]:L["kernel_off"] + n] = k.ravel()
                p[L["bias_off"]:L["bias_off"] + c] = np.asarray(next(it), np.float32)
                if L["has_bn"]:
                    p[L["gamma_off"]:L["gamma_off"] + c] = np.asarray(next(it), np.float32)
                    p[L["beta_off"]:L["beta_off"] + c] = np.asarray(next(it), np.float32)
                    s[L["moving_mean_off"]:L["moving_mean_off"] + c] = np.asarray(next(it), np.float32)
                    s[L["moving_var_off"]:L["moving_var_off"] + c] = np.asarray(next(it), np.float32)
        except StopIteration:
            raise OctError("set_weights: too few arrays") from None
        if next(it, None) is not None:
            raise OctError("set_weights: too many arrays")
        self.params.copy_(torch.from_numpy(p)); self.state.copy_(torch.from_numpy(s))

    def debug_bn_record(self, layer: int) -> torch.Tensor:
        """The (9, cout) f32 BN record of ``layer``: rows a, b, mean, rstd (consumers apply relu(a*z + b) on load), c1, c2
        (BN-backward means) and ga, gb, gd (the BN-backward transform dz = ga*g' + gb*z + gd)."""
        ptr = _hip.lib().oct_unet_debug_activation(self._h, layer, 2)
        if not ptr:
            raise OctError("no BN record for this layer")
        c = self.layers[layer]["cout"]
        off = ptr - self.workspace.data_ptr()
        return self.workspace[off:off + 4 * 9 * c].view(torch.float32).view(9, c)

    def debug_layer_fused(self, layer: int) -> bool:
        """True if, in the last backward, the layer's BN-backward transform was applied on load by its consumers: its
        gradient buffer (``debug_activation(layer, 1)``) then holds the masked gradient g', not dz."""
        r = _hip.lib().oct_unet_debug_layer_fused(self._h, layer)
        if r < 0:
            raise OctError("no such layer")
        return bool(r)

    def debug_dz(self, layer: int) -> torch.Tensor:
        """dz of ``layer`` after a backward, float64: the gradient buffer itself where the stand-alone BN-backward pass
        ran, else the transform of the stored g' and z with the record's rows (what the consumers formed on load)."""
        g = self.debug_activation(layer, 1).double()
        if not self.debug_layer_fused(layer):
            return g
        rec = self.debug_bn_record(layer).double()
        return rec[6] * g + (rec[7] * self.debug_activation(layer, 0).double() + rec[8])

    def set_option(self, name: str, value: int) -> None:
        """Edit this handle's copy of a tuning / arithmetic option (``_hip.set_option`` edits the defaults new handles copy)."""
        _hip.check(_hip.lib().oct_unet_set_option(self._h, name.encode(), int(value)), f"oct_unet_set_option({name})")

    def handle_option(self, name: str) -> int:
        """The value of a tuning option as this handle snapshotted it at creation."""
        import ctypes as C
        v = C.c_int(0)
        _hip.check(_hip.lib().oct_unet_get_option(self._h, name.encode(), C.byref(v)), f"oct_unet_get_option({name})")
        return int(v.value)

    def debug_activation(self, layer: int, which: int = 0) -> torch.Tensor:
        """A layer's saved pre-BN output (which=0) or gradient buffer (which=1), max_batch-sized; a float32 view in
        f32 mode, a float32 COPY of the bf16 storage in bf16 mode."""
        ptr = _hip.lib().oct_unet_debug_activation(self._h, layer, which)
        if not ptr:
            raise OctError("no such activation")
        L = self.layers[layer]
        n = self.cfg.max_batch * L["out_h"] * L["out_w"] * L["cout"]
        off = ptr - self.workspace.data_ptr()
        shape = (self.cfg.max_batch, L["out_h"], L["out_w"], L["cout"])
        if self.cfg.dtype == 1:
            return self.workspace[off:off + 2 * n].view(torch.bfloat16).view(shape).float()
        return self.workspace[off:off + 4 * n].view(torch.float32).view(shape)
