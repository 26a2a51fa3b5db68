"""Data-parallel plumbing: one process per GPU, ``torch.distributed`` (backend "nccl" = RCCL over xGMI on
ROCm; "gloo" on CPU for tests).  Mirrors what the reference gets from ``tf.distribute.MirroredStrategy``
(training/training.py:185-188,243): the global batch is split across replicas, BatchNorm statistics stay
per replica, the loss is scaled by 1/num_replicas and the gradients are sum-all-reduced.

The exchange step is ONE all-reduce of the engine's flat fp32 gradient buffer (487 403 floats = 1.95 MB for
the default U-Net): a latency-bound message on xGMI, so it is never bucketed further."""
from __future__ import annotations

import os
from typing import Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist


def env_rank() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torchrun environment; (0, 0, 1) when not launched by it."""
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


_forced = False      # a process group exists although world_size == 1 (RCCL rehearsal on one GPU)


def init(backend: Optional[str] = None, force: bool = False) -> Tuple[int, int, int]:
    """Join the process group named by the torchrun environment.  ``force`` (or OCT_FORCE_COLLECTIVE=1): create the
    group even when WORLD_SIZE is 1 -- a one-rank RCCL communicator, so that the exchange step of a training step
    (``GradReducer``: tail event -> side-stream all-reduce -> encoder-segment all-reduce -> join) runs through RCCL on
    a single GPU exactly as it does on eight."""
    global _forced
    rank, local_rank, world = env_rank()
    force = force or os.environ.get("OCT_FORCE_COLLECTIVE") == "1"
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
        _forced = world == 1
    return rank, local_rank, world


def shutdown() -> None:
    global _forced
    if dist.is_initialized():
        dist.destroy_process_group()
    _forced = False


def collective_active() -> bool:
    """True when gradient exchange goes through ``torch.distributed``: more than one rank, or a forced one-rank group."""
    return dist.is_initialized() and (dist.get_world_size() > 1 or _forced)


def describe() -> dict:
    """What the collective library itself reports (for the bench record): backend, world size, and every rank's device."""
    if not dist.is_initialized():
        return {"backend": None, "world_size": 1, "ranks": []}
    me = {"rank": dist.get_rank(), "local_rank": env_rank()[1]}
    if torch.cuda.is_available():
        i = torch.cuda.current_device()
        pr = torch.cuda.get_device_properties(i)
        me.update(device=f"cuda:{i}", gpu=pr.name, uuid=str(getattr(pr, "uuid", "")), pci_bus_id=getattr(pr, "pci_bus_id", None))
    ranks = [None] * dist.get_world_size()
    dist.all_gather_object(ranks, me)
    return {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "ranks": ranks}


def world_size() -> int:
    return dist.get_world_size() if dist.is_initialized() else 1


def shard_batch(global_batch: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [start, stop) slice of a global batch for this replica (Keras splits the generator's
    GLOBAL batch across replicas).  Requires equal shards so that the mean of per-replica macro-Dice losses
    equals the global macro-Dice loss."""
    if global_batch % world:
        raise ValueError(f"global batch {global_batch} is not divisible by {world} replicas")
    per = global_batch // world
    return rank * per, (rank + 1) * per


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous index range of an n-item evaluation set for this rank (sizes differ by at most one)."""
    base, extra = divmod(n, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def allreduce_gradients(grads: torch.Tensor) -> None:
    """Sum-all-reduce of the flat gradient buffer, in place (gradients were pre-scaled by 1/world)."""
    if world_size() > 1:
        dist.all_reduce(grads, op=dist.ReduceOp.SUM)


def broadcast_parameters(params: torch.Tensor, state: torch.Tensor, src: int = 0) -> None:
    if world_size() > 1:
        dist.broadcast(params, src)
        dist.broadcast(state, src)


def average_moving_stats(state: torch.Tensor) -> torch.Tensor:
    """BN moving statistics are per replica during training; they are mean-reduced only when read
    (checkpoint / validation), as MirroredStrategy does for ON_READ variables."""
    out = state.clone()
    if world_size() > 1:
        dist.all_reduce(out, op=dist.ReduceOp.SUM)
        out /= world_size()
    return out


def max_over_ranks(value: float, device=None) -> float:
    if world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier() -> None:
    if world_size() > 1:
        dist.barrier()


def broadcast_object(obj, src: int = 0):
    """The same small Python object on every rank (rank ``src``'s copy)."""
    if world_size() == 1:
        return obj
    box = [obj]
    dist.broadcast_object_list(box, src=src)
    return box[0]


def shared_seed(seed: Optional[int]) -> int:
    """One shuffle / augmentation seed for ALL ranks: every rank must draw the same global batches so that the
    per-rank slices of ``shard_batch`` partition ONE batch (MirroredStrategy splits one generator's batch).
    ``seed=None`` (the reference's unseeded shuffle) draws from OS entropy on rank 0 and broadcasts it."""
    if seed is None:
        seed = int(np.random.SeedSequence().generate_state(1)[0]) if env_rank()[0] == 0 or world_size() == 1 else 0
    return int(broadcast_object(int(seed)))


class GradReducer:
    """The exchange step of one training step: ``backward`` + sum-all-reduce of the flat gradient buffer.

    ``overlap=True``: the engine records an event as soon as the bottleneck + decoder + head gradients
    (the tail segment of the buffer, 97 % of the floats at the default config) are final; that segment is all-reduced
    on a side stream behind the event while the encoder backward still runs on the compute stream, the small encoder
    segment after backward.  Two collectives instead of one, the large one hidden.  ``overlap=False``: ONE all-reduce
    of the whole buffer after backward on the compute stream."""

    def __init__(self, engine, overlap: bool = True):
        self.engine = engine
        self.active = collective_active()
        self.overlap = bool(overlap) and self.active
        if self.overlap:
            self.side = torch.cuda.Stream(device=engine.device)
            self.event = torch.cuda.Event()
            self.off = engine.grad_tail_offset()
            engine.set_tail_event(self.event)

    def backward_and_reduce(self, labels: torch.Tensor, macro: bool = True, loss_scale: float = 1.0, timed: bool = False):
        """``timed``: return [(start, end) event pairs] of the collectives issued (tail segment first), else None."""
        eng = self.engine
        eng.backward(labels, macro=macro, loss_scale=loss_scale)
        if not self.active:
            return None
        main = torch.cuda.current_stream(eng.device)

        def ev(stream):
            e = torch.cuda.Event(enable_timing=True); e.record(stream); return e
        pairs = []
        if not self.overlap:
            a = ev(main) if timed else None
            dist.all_reduce(eng.grads, op=dist.ReduceOp.SUM)
            if timed:
                pairs.append((a, ev(main)))
            return pairs if timed else None
        self.side.wait_event(self.event)            # GPU-side: the tail segment is final
        with torch.cuda.stream(self.side):
            a = ev(self.side) if timed else None
            dist.all_reduce(eng.grads[self.off:], op=dist.ReduceOp.SUM)
            if timed:
                pairs.append((a, ev(self.side)))
        a = ev(main) if timed else None
        dist.all_reduce(eng.grads[:self.off], op=dist.ReduceOp.SUM)      # encoder segment, after the whole backward
        if timed:
            pairs.append((a, ev(main)))
        main.wait_stream(self.side)                 # the optimizer step needs both
        return pairs if timed else None

    def close(self) -> None:
        if self.overlap:
            self.engine.set_tail_event(None)
            self.overlap = False
