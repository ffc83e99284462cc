"""Whole-step HIP graph: forward + backward + (clip + AdamW) captured once, replayed every step.

The path is ~1,500 short kernel launches per step; issued one by one from Python the host, not the GPU, sets the step
time.  The MI355X-first answer (task brief: "HIP streams and graphs instead of a tracing compiler") is to capture the
step once and replay it: no per-launch host cost, and independent branches of the step (the two encoders, weight-gradient
GEMMs) become parallel branches of the graph, filling CUs that a single short GEMM leaves idle during its cold start and
its C-tile stores.

What makes a replay a real training step and not a re-run of the captured one:
  * inputs are copied into static device buffers before each replay;
  * dropout / router-noise keys are INDIRECT seeds resolved from a device epoch word that the graph itself advances
    (csrc/common.h ``resolve_seed``), so every replay draws fresh masks and backward regenerates the forward's;
  * the optimiser's learning rate and step count live in device memory and are advanced inside the graph
    (``FusedAdamW.make_capturable``), so bias correction and schedules follow the real step number.
Data-dependent host decisions cannot be captured: the MoE layers switch to their dense dispatch (every expert on every
token, combined with the routing weights -- exact, and free at one token per sample; see modeling/moe/moe_layer.py), and an
expert no token chose is skipped by the optimiser through a device-side routed-token count instead of a ``grad is None``.  With data parallelism the gradient exchange stays outside the graph (forward+backward is one
graph, the all-reduce is launched eagerly, the optimiser step is a second graph).
"""

from typing import Callable, Dict, Optional

import torch

from .hip import blocks as _blocks


class GraphedTrainStep:
    def __init__(self, model: torch.nn.Module, optimizer, batch: Dict[str, torch.Tensor], *, loss_of: Optional[Callable] = None,
                 reducer=None, warmup: int = 3, parallel_towers: bool = True,
                 capture_error_mode: str = 'global', capture_stream=None, defer_wgrad: bool = True):
        """``batch``: keyword tensors of ``model.forward`` (shapes are fixed by the capture).
        Construct this BEFORE training the model eagerly on the default stream (or run such steps under
        ``torch.cuda.stream(side_stream)``): autograd binds each parameter's gradient-accumulation node to the stream of its
        first backward, and a node bound to the legacy default stream cannot be joined into a capture ("capturing stream has
        unjoined work") -- the same rule as PyTorch's whole-network capture recipe; the warm-up here runs on a side stream.  ``loss_of(output)`` picks the
        scalar to differentiate (default ``output.loss``).  ``reducer``: a ``dp.GradReducer`` in NON-overlap mode.
        ``parallel_towers``: the vision encoder runs as a parallel branch of the graph (measured on MI355X, cfg2, B=32:
        13.7 -> 10.6 ms/step).  ``defer_wgrad``: the weight-gradient GEMMs of both encoders are issued (grouped) after
        backward returned, where they run alone at full efficiency."""
        self.model, self.opt, self.reducer = model, optimizer, reducer
        self.loss_of = loss_of or (lambda out: out.loss)
        self.static = {k: v.clone() for k, v in batch.items()}
        dev = next(iter(self.static.values())).device
        _blocks.enable_indirect_seeds(dev)
        if parallel_towers and hasattr(model, 'encode_visual'):
            model.parallel_towers = True
        for m in model.modules():                      # MoE layers: dispatch without the host read of the routing counts
            if hasattr(m, 'enable_dense_dispatch'):
                m.enable_dense_dispatch(True)
        self._defer_wgrad = defer_wgrad
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):                      # warm-up off the default stream (allocator pools, lazy tables, tile attributes)
            for _ in range(max(1, warmup)):    # at least one eager step in the capture's own configuration (streams, dense MoE dispatch)
                self._fwd_bwd()
                if reducer is not None:
                    reducer.reduce()
                self.opt.step()
        cur.wait_stream(side)
        torch.cuda.synchronize()
        if hasattr(self.opt, 'make_capturable'):
            self.opt.make_capturable(dev)
        self.g_main = torch.cuda.CUDAGraph()
        self.g_opt = None
        # capture_error_mode 'thread_local' when other threads may touch the device during the capture (the process group's
        # watchdog polls events)
        if reducer is None:
            with torch.cuda.graph(self.g_main, capture_error_mode=capture_error_mode, stream=capture_stream):
                self.loss = self._fwd_bwd()
                self.opt.step()
        else:
            with torch.cuda.graph(self.g_main, capture_error_mode=capture_error_mode):
                self.loss = self._fwd_bwd()
            # the gradients now have their final, static addresses: the reducer finds the arenas it will all-reduce in place
            # after every replay (and packs the few stand-alone gradients, re-pointing p.grad) before the optimiser graph
            # is captured against those addresses
            reducer.prepare_static()
            self.g_opt = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g_opt, pool=self.g_main.pool(), capture_error_mode=capture_error_mode):
                self.opt.step()

    def _fwd_bwd(self):
        from .hip import kernels as K
        _blocks.advance_rng_epoch()
        self.opt.zero_grad(set_to_none=True)
        prev_defer, K.WGRAD_DEFER_TO_STEP_END = K.WGRAD_DEFER_TO_STEP_END, self._defer_wgrad
        try:
            out = self.model(**self.static)
            loss = self.loss_of(out)
            # fp16 mode: the optimiser's device-side loss scale (GradScaler's role) multiplies the loss inside the capture
            (self.opt.scale_loss(loss) if hasattr(self.opt, 'scale_loss') else loss).backward()
            K.wgrad_flush_all()
        finally:
            K.WGRAD_DEFER_TO_STEP_END = prev_defer
        return loss.detach()

    def __call__(self, batch: Optional[Dict[str, torch.Tensor]] = None) -> torch.Tensor:
        if batch is not None:
            for k, v in batch.items():
                self.static[k].copy_(v, non_blocking=True)
        self.g_main.replay()
        if self.g_opt is not None:
            self.reducer.reduce_static()
            for m in self.model.modules():              # an expert is active when ANY rank routed a token to it
                a = getattr(m, '_active', None)
                if a is not None and getattr(m, 'dense_dispatch', False):
                    torch.distributed.all_reduce(a)
            self.g_opt.replay()
        if hasattr(self.opt, 'note_replays'):
            self.opt.note_replays(1)
        return self.loss
