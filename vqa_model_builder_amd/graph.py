"""Whole-step HIP graphs: forward + backward + (clip + AdamW) captured once, replayed every step.

The path is ~530 short kernel launches per step; issued one by one from Python the host, not the GPU, sets the step
time.  The MI355X-first answer (task brief: "HIP streams and graphs instead of a tracing compiler") is to capture the
step once and replay it: no per-launch host cost, and independent branches of the step (the two encoders) become
parallel branches of the graph, filling CUs that a single short GEMM leaves idle during its cold start and its C-tile stores.

What makes a replay a real training step and not a re-run of the captured one:
  * inputs are copied into static device buffers before each replay;
  * dropout / router-noise keys are INDIRECT seeds resolved from a device epoch word that the graph itself advances
    (csrc/common.h ``resolve_seed``), so every replay draws fresh masks and backward regenerates the forward's;
  * the optimiser's learning rate, step count and (fp16 mode) loss scale live in device memory and are advanced inside the
    graph (``FusedAdamW.make_capturable``), so bias correction, schedules and the GradScaler policy follow the real step.
Data-dependent host decisions cannot be captured: the MoE layers switch to their dense dispatch (every expert on every
token, combined with the routing weights -- exact, and free at one token per sample; see modeling/moe/moe_layer.py), and an
expert no token chose is skipped by the optimiser through a device-side routed-token count instead of a ``grad is None``.

ONE GPU: one graph for the whole step (two parallel encoder branches, all weight-gradient GEMMs grouped at the end).

DATA PARALLEL (a ``dp.GradReducer`` is given): collectives stay outside captures, so the step is cut where the gradient
exchange can start -- the autograd graph is severed at the encoder outputs and the step becomes a chain of graphs

    F          both encoders forward (parallel branches)
    H          fusion + MoE + answer head forward AND backward  -> gradients of the head / fusion / MoE and of the encoder outputs
    B1 .. B4   the encoders' backward cut BY DEPTH (``dp_split='depth'``, default): segment j holds the text AND the vision layers of
               one quarter of the depth as parallel branches of one graph (resumable runner backward: blocks.py ``backward_steps``);
               the last one also the embeddings.  (``dp_split='towers'``, the first form of round 2: T, T2 = text backward upper / lower
               half, then V, V2 = vision backward -- one tower at a time: 8.8 - 9.0 ms per step against 8.4 - 8.5 ms by depth, same box.)
    O          clip + AdamW (+ loss-scale update)

and the host replays  F, H, [all-reduce H], B1, [all-reduce B1], ..., B4, [all-reduce B4], wait, O:  every ``all_reduce`` is
asynchronous on RCCL's own stream, IN PLACE on the contiguous runs of the segment's gradients inside the blocks' arenas, ordered after
the graph that produced them and running beside the next graph, so only the LAST segment's exchange (a quarter of the encoders: 97 MB
fp32 / 48 MB with bf16 buckets, plus the gathered embedding rows) is exposed -- round 1 exposed all 0.98 GB between one backward graph
and the optimiser graph.  ``comm_stats()`` reports the measured exposed time and the GPU time of every graph; tests/test_dp_gpu.py
checks the captured step against the eager data-parallel step (both splits, fp32 and bf16 buckets, MoE, and over a one-rank RCCL
group).  Forks inside a capture are ONE level deep (a fork nested in a forked branch faults in this runtime:
profiles/r02/nested_fork_capture.md): in B2 .. B4 the vision tower's resume runs on the tower side stream, forked from and joined to
the capture stream directly.
"""

from typing import Callable, Dict, Optional

import torch

from .hip import blocks as _blocks


class GraphedTrainStep:
    def __init__(self, model: torch.nn.Module, optimizer, batch: Dict[str, torch.Tensor], *, loss_of: Optional[Callable] = None,
                 reducer=None, warmup: int = 3, parallel_towers: bool = True,
                 capture_error_mode: str = 'global', capture_stream=None, defer_wgrad: bool = True, segmented: Optional[bool] = None,
                 moe_branches: int = 1, split_encoders: bool = True, sparse_embeddings: bool = True, dp_split: str = 'depth',
                 exchange_on_side_stream: bool = True, wire_optimizer: bool = True, dp_segments: str = 'tapered'):
        """``batch``: keyword tensors of ``model.forward`` (shapes are fixed by the capture).
        Construct this BEFORE training the model eagerly on the default stream (or run such steps under
        ``torch.cuda.stream(side_stream)``): autograd binds each parameter's gradient-accumulation node to the stream of its
        first backward, and a node bound to the legacy default stream cannot be joined into a capture ("capturing stream has
        unjoined work") -- the same rule as PyTorch's whole-network capture recipe; the warm-up here runs on a side stream.
        ``loss_of(output)`` picks the scalar to differentiate (default ``output.loss``).  ``reducer``: a ``dp.GradReducer``
        without hooks attached.  ``parallel_towers``: the vision encoder runs as a parallel branch (measured on MI355X, cfg2,
        B=32: 13.7 -> 10.6 ms/step).  ``defer_wgrad``: weight-gradient GEMMs are issued grouped at the end of their graph.
        ``sparse_embeddings`` (segmented step): the word-embedding gradient is exchanged as gathered (ids, rows) (dp.prepare_static).
        ``split_encoders`` (segmented step): each encoder's backward as two graphs (upper / lower half of its layers).
        ``dp_split`` (segmented step): 'depth' (default) cuts BOTH encoders' backward by depth into up to four graphs B1 .. B4, each
        holding the text and the vision layers of that depth as parallel branches (the two towers keep filling each other's gaps, as in
        the one-graph step); 'towers' is the first form of this round: text backward (T, T2) then vision backward (V, V2), one tower at a time.
        ``dp_segments`` (depth split): 'tapered' (default) = a one-layer last segment, 'even' = equal segments.
        ``exchange_on_side_stream`` (segmented step): each segment's pack / all-reduce / unpack chain runs on a stream of its own.
        ``wire_optimizer`` (segmented step, bf16 buckets, FusedAdamW): the optimiser reads the summed bfloat16 gradients in the exchange's
        staging buffers; ``p.grad`` keeps the rank's local gradient.
        ``moe_branches``: MoE experts on side streams = parallel branches of the capture (0 off, 1 the specialised experts, 2 all).
        ``segmented`` (default: with a reducer whose world > 1 and a model that offers ``encode_both`` /
        ``forward_from_features``): the multi-graph data-parallel step described in the module docstring."""
        self.model, self.opt, self.reducer = model, optimizer, reducer
        self.loss_of = loss_of or (lambda out: out.loss)
        self.static = {k: v.clone() for k, v in batch.items()}
        dev = next(iter(self.static.values())).device
        _blocks.enable_indirect_seeds(dev)
        if parallel_towers and (hasattr(model, 'encode_visual') or hasattr(model, 'encode_both')):
            model.parallel_towers = True
        for m in model.modules():                      # MoE layers: dispatch without the host read of the routing counts
            if hasattr(m, 'enable_dense_dispatch'):
                m.enable_dense_dispatch(True)
                m.parallel_branches = moe_branches            # specialised experts as a parallel branch of the graph (cfg3: 11.1 -> 10.4 ms)
        self._defer_wgrad = defer_wgrad
        if reducer is None and hasattr(optimizer, 'fuse_wgrad_norm'):
            # single GPU: the clipping norm's weight-gradient share is summed by the GEMMs that store those gradients (one backward per step
            # here, nothing all-reduced behind it): -0.98 GB of re-reads per cfg2 step
            optimizer.fuse_wgrad_norm(True, dev)
        elif hasattr(optimizer, 'fuse_wgrad_norm'):
            optimizer.fuse_wgrad_norm(False)                # data parallel: the norm must be the all-reduced gradients'
        can_segment = hasattr(model, 'encode_both') and hasattr(model, 'forward_from_features')
        if segmented is None:
            segmented = reducer is not None and not getattr(reducer, 'single', True) and can_segment
        self.segmented = bool(segmented and can_segment and reducer is not None)
        # every encoder's backward is cut once more (upper / lower half of its layers): the upper half's arena leaves while the
        # lower half computes, and only half of the LAST encoder's arena is left to travel when the step's compute is done
        self._order, self._splits, self._depth = ('H', 'T', 'V'), {}, None
        tb = getattr(getattr(model, 'text_encoder', None), 'encoder', None)
        vb = getattr(getattr(model, 'visual_encoder', None), 'backbone', None)
        if (self.segmented and split_encoders and dp_split == 'depth' and all(b is not None and hasattr(b, 'resume_backward') for b in (tb, vb))
                and min(tb.config.num_hidden_layers, vb.config.num_hidden_layers) >= 2 and getattr(model, 'parallel_towers', False)):
            # depth-wise segments: segment j of n holds layers [cut[j], cut[j-1]) of each tower (descending), the last one the embeddings too
            n = min(4, tb.config.num_hidden_layers, vb.config.num_hidden_layers)

            def cuts_of(L):
                # descending layer boundaries.  'tapered' (default): the LAST segment is one layer (+ the embeddings) -- its exchange is the one
                # nothing can hide, so it is made the smallest (28 MB of bf16 per layer pair instead of 85 MB) -- and the other L - 1 layers are
                # dealt evenly over the first n - 1 segments; 'even': L / n layers each (round 2)
                if dp_segments == 'even' or n == 1:
                    return [L - (L * j) // n for j in range(1, n)]
                sizes = [(L - 1) // (n - 1) + (1 if j < (L - 1) % (n - 1) else 0) for j in range(n - 1)]
                out, at = [], L
                for sz in sizes:
                    at -= sz
                    out.append(at)
                return out
            self._depth = {'n': n, 'blocks': {'T': tb, 'V': vb}, 'cuts': {k: cuts_of(b.config.num_hidden_layers) for k, b in (('T', tb), ('V', vb))}}
            self._order = ('H',) + tuple(f'B{j + 1}' for j in range(n))
        elif self.segmented and split_encoders:
            order = ['H']
            for seg, blk in (('T', getattr(getattr(model, 'text_encoder', None), 'encoder', None)),
                             ('V', getattr(getattr(model, 'visual_encoder', None), 'backbone', None))):
                order.append(seg)
                if blk is not None and hasattr(blk, 'resume_backward') and blk.config.num_hidden_layers >= 2:
                    self._splits[seg] = (blk, blk.config.num_hidden_layers // 2)
                    order.append(seg + '2')
            self._order = tuple(order)
        self._exposed_ms, self._comm_events, self._replays = [], None, 0
        self._exchange_stream = torch.cuda.Stream() if (self.segmented and exchange_on_side_stream) else None
        self._segment_marks = []
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):                      # warm-up off the default stream (allocator pools, lazy tables, tile attributes)
            for _ in range(max(1, warmup)):    # at least one eager step in the capture's own configuration (streams, dense MoE dispatch)
                if self.segmented:
                    self._segment_F()
                    for name in self._order:
                        self._segment_fn(name)()
                else:
                    self._fwd_bwd()
                if reducer is not None:
                    reducer.reduce()
                    self._sync_routed_counts()
                self.opt.step()
        cur.wait_stream(side)
        torch.cuda.synchronize()
        if hasattr(self.opt, 'make_capturable'):
            self.opt.make_capturable(dev)
        self.g_main = torch.cuda.CUDAGraph()
        self.g_opt = None
        self.graphs = {}
        # capture_error_mode 'thread_local' when other threads may touch the device during the capture (the process group's
        # watchdog polls events)
        if reducer is None:
            with torch.cuda.graph(self.g_main, capture_error_mode=capture_error_mode, stream=capture_stream):
                self.loss = self._fwd_bwd()
                self.opt.step()
        elif not self.segmented:
            with torch.cuda.graph(self.g_main, capture_error_mode=capture_error_mode):
                self.loss = self._fwd_bwd()
            # the gradients now have their final, static addresses: the reducer finds the arenas it will all-reduce in place
            # after every replay (and packs the few stand-alone gradients, re-pointing p.grad) before the optimiser graph
            # is captured against those addresses
            reducer.prepare_static()
            self.g_opt = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g_opt, pool=self.g_main.pool(), capture_error_mode=capture_error_mode):
                self.opt.step()
        else:
            kw = dict(capture_error_mode=capture_error_mode)
            with torch.cuda.graph(self.g_main, **kw):                      # F
                self._segment_F()
            pool = self.g_main.pool()
            order = self._order
            for name in order:
                fn = self._segment_fn(name)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=pool, **kw):
                    fn()
                self.graphs[name] = g
            seg_of = {}
            for n, p in model.named_parameters():
                seg_of[id(p)] = self._segment_of(n)
            sparse = []
            enc = getattr(getattr(model, 'text_encoder', None), 'encoder', None)
            if sparse_embeddings and hasattr(enc, 'sparse_grad_rows'):
                sparse.append(enc.sparse_grad_rows)           # the word table travels as (ids, rows), not as a 196-MB dense gradient
            self._check_split_gradients_alias_their_arena()
            reducer.prepare_static(seg_of, order, sparse=sparse)
            # the device-side packing of each segment (stand-alone gradients -> pack buffer, bf16 wire copies) is a small graph of
            # its own, replayed right behind the segment's backward graph
            for name in order:
                if not reducer.segment_needs_pack(name):
                    self.graphs['pack' + name] = None     # fp32 buckets, every gradient inside an arena: nothing to do (an EMPTY capture is what
                    continue                              # printed "The CUDA Graph is empty" in round 2's bench log: the dp_model leg's fp32 segments)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=pool, **kw):
                    reducer.pack_segment(name)
                self.graphs['pack' + name] = g
            if wire_optimizer and getattr(reducer, 'grad_dtype', 'fp32') == 'bf16' and hasattr(self.opt, 'wire_grads') and not getattr(reducer, 'average', False):
                # bf16 buckets: the optimiser reads the all-reduced bfloat16 sums where RCCL leaves them (no copy back into the fp32 arenas)
                self.opt.wire_grads = reducer.wire_gradient_ptrs()
                reducer.consume_on_wire(True)
            self.g_opt = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g_opt, pool=pool, **kw):
                self.opt.step()
            self._ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]

    def _check_split_gradients_alias_their_arena(self):
        """A block whose backward is cut over several graphs hands autograd views of its gradient arena BEFORE the lower layers' slots are
        written (a later graph fills them).  That is correct only if ``p.grad`` is that very view; had AccumulateGrad cloned one (a
        layout-contract or refcount miss, a non-contiguous parameter), ``p.grad`` would be private storage holding not-yet-computed
        data and the exchange would pack garbage every step, silently.  Checked once, after the capture."""
        blocks = list(self._depth['blocks'].values()) if self._depth is not None else [b for b, _ in self._splits.values()]
        for blk in blocks:
            arena = getattr(blk, '_split_arena_ptr', None)
            if arena is None:
                continue
            for key, p in blk._flat:
                if p.requires_grad and p.grad is not None and p.grad.untyped_storage().data_ptr() != arena:
                    raise RuntimeError(f'segmented step: the gradient of {type(blk).__name__} parameter {key!r} does not alias the block\'s gradient arena '
                                       '(autograd copied the view it was handed): its lower-layer slots would never reach p.grad')

    # ---- single-graph step -----------------------------------------------------------------------------------------------
    def _backward(self, loss):
        # fp16 mode: the optimiser's device-side loss scale (GradScaler's role) multiplies the loss inside the capture
        (self.opt.scale_loss(loss) if hasattr(self.opt, 'scale_loss') else loss).backward()

    def _fwd_bwd(self):
        from .hip import kernels as K
        _blocks.advance_rng_epoch()
        self.opt.zero_grad(set_to_none=True)
        prev_defer, K.WGRAD_DEFER_TO_STEP_END = K.WGRAD_DEFER_TO_STEP_END, self._defer_wgrad
        try:
            K.prefetch_begin()                          # the weight-prefetch stream: a one-level fork of the capture stream (kernels.py)
            out = self.model(**self.static)
            loss = self.loss_of(out)
            self._backward(loss)
            K.prefetch_join()
            K.wgrad_flush_all()
        finally:
            K.WGRAD_DEFER_TO_STEP_END = prev_defer
        return loss.detach()

    # ---- the data-parallel segments --------------------------------------------------------------------------------------
    def _segment_F(self):
        _blocks.advance_rng_epoch()
        self.opt.zero_grad(set_to_none=True)
        s = self.static
        from .hip import kernels as K
        K.prefetch_begin()
        self._enc = self.model.encode_both(s['pixel_values'], s['input_ids'], s['attention_mask'])
        K.prefetch_join()

    def _segment_H(self):
        from .hip import kernels as K
        s = self.static
        self._cut = [t.detach().requires_grad_(True) for t in self._enc]          # sever the autograd graph at the encoder outputs
        prev_defer, K.WGRAD_DEFER_TO_STEP_END = K.WGRAD_DEFER_TO_STEP_END, self._defer_wgrad
        try:
            out = self.model.forward_from_features(*self._cut, s['attention_mask'], s.get('labels'))
            loss = self.loss_of(out)
            self._backward(loss)
            K.wgrad_flush_all()
        finally:
            K.WGRAD_DEFER_TO_STEP_END = prev_defer
        self.loss = loss.detach()

    def _encoder_backward(self, idx):
        from .hip import kernels as K
        outs = [self._enc[i] for i in idx]
        grads = [self._cut[i].grad for i in idx]
        pairs = [(o, g) for o, g in zip(outs, grads) if g is not None and o.requires_grad]
        prev_defer, K.WGRAD_DEFER_TO_STEP_END = K.WGRAD_DEFER_TO_STEP_END, self._defer_wgrad
        try:
            K.prefetch_begin()
            if pairs:
                torch.autograd.backward([o for o, _ in pairs], [g for _, g in pairs])
            K.prefetch_join()
            K.wgrad_flush_all()
        finally:
            K.WGRAD_DEFER_TO_STEP_END = prev_defer

    def _split_backward(self, seg, idx):
        blk, at = self._splits.get(seg, (None, None))
        if blk is not None:
            blk.split_backward_after = at               # the autograd node stops after the upper half of the layers ...
        try:
            self._encoder_backward(idx)
        finally:
            if blk is not None:
                blk.split_backward_after = None

    def _resume_backward(self, seg):
        from .hip import kernels as K
        prev_defer, K.WGRAD_DEFER_TO_STEP_END = K.WGRAD_DEFER_TO_STEP_END, self._defer_wgrad
        try:
            K.prefetch_begin()
            self._splits[seg][0].resume_backward()      # ... and the lower half + embeddings run here
            K.prefetch_join()
            K.wgrad_flush_all()
        finally:
            K.WGRAD_DEFER_TO_STEP_END = prev_defer

    def _segment_fn(self, name):
        if self._depth is not None and name.startswith('B'):
            j = int(name[1:]) - 1
            return (lambda: self._depth_first()) if j == 0 else (lambda: self._depth_resume(last=(j == self._depth['n'] - 1)))
        return getattr(self, '_segment_' + name)

    def _depth_first(self):
        """B1: both encoders' autograd nodes in ONE backward call -- the vision node runs on the side stream its forward ran on (a parallel
        branch of the capture), each node stops after its top segment of layers (split_backward_after) and hands out the gradient views."""
        d = self._depth
        for k, blk in d['blocks'].items():
            blk.split_backward_after = tuple(d['cuts'][k])
        try:
            self._encoder_backward((0, 1, 2, 3))
        finally:
            for blk in d['blocks'].values():
                blk.split_backward_after = None

    def _depth_resume(self, last):
        """B2 ..: the next segment of layers of both towers, vision on the tower side stream (one-level fork, joined before the grouped
        weight-gradient launches)."""
        from .hip import kernels as K
        d = self._depth
        cur = torch.cuda.current_stream()
        side = getattr(self.model, '_tower_stream', None) or torch.cuda.Stream()
        prev_defer, K.WGRAD_DEFER_TO_STEP_END = K.WGRAD_DEFER_TO_STEP_END, self._defer_wgrad
        try:
            K.prefetch_begin()
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                d['blocks']['V'].resume_backward(to_end=last)
            d['blocks']['T'].resume_backward(to_end=last)
            cur.wait_stream(side)
            K.prefetch_join()
            K.wgrad_flush_all()
        finally:
            K.WGRAD_DEFER_TO_STEP_END = prev_defer

    def _segment_T(self):
        self._split_backward('T', (2, 3))            # (text_pooled, text_sequence)

    def _segment_T2(self):
        self._resume_backward('T')

    def _segment_V(self):
        self._split_backward('V', (0, 1))            # (visual_pooled, visual_spatial)

    def _segment_V2(self):
        self._resume_backward('V')

    def _segment_of(self, name):
        import re
        seg = 'V' if name.startswith('visual_encoder.') else 'T' if name.startswith('text_encoder.') else 'H'
        if self._depth is not None and seg != 'H':
            m = re.search(r'\.layers?\.(\d+)\.', name)
            cuts = self._depth['cuts'][seg]                 # descending layer indices: segment j holds layers >= cuts[j] (and < cuts[j-1])
            if m is None:
                return f"B{self._depth['n']}"               # embeddings, pre-LayerNorm, patch projection: the last segment
            l = int(m.group(1))
            for j, c in enumerate(cuts):
                if l >= c:
                    return f'B{j + 1}'
            return f"B{self._depth['n']}"
        if seg in self._splits:
            m = re.search(r'\.layers?\.(\d+)\.', name)
            if m is None or int(m.group(1)) < self._splits[seg][1]:
                seg += '2'                              # lower layers, embeddings, projections
        return seg

    def describe(self) -> str:
        if self.reducer is None:
            return 'hip-graph (one graph: 2 parallel encoder branches, grouped weight gradients, clip + AdamW)'
        if not self.segmented:
            return 'hip-graph forward+backward, eager all-reduce, hip-graph optimiser'
        if self._depth is not None:
            return (f'{len(self._order) + 2} hip-graphs (encoders fwd | fusion+head fwd+bwd | {self._depth["n"]} depth segments of BOTH encoders\' backward, text and vision as parallel '
                    f'branches | optimiser); each segment\'s gradient runs all-reduced ({self.reducer.grad_dtype}) in place beside the next segment\'s graph')
        return (f'{len(self._order) + 2} hip-graphs (encoders fwd | fusion+head fwd+bwd | text bwd | vision bwd; encoder backwards cut in {"two" if self._splits else "one"} | optimiser); each block\'s gradient arena '
                f'all-reduced ({self.reducer.grad_dtype}) beside the next block\'s graph')

    def comm_stats(self) -> Dict[str, float]:
        if not self._exposed_ms:
            return {}
        self._exposed_ms = [e if isinstance(e, float) else e[0].elapsed_time(e[1]) for e in self._exposed_ms]
        v = sorted(self._exposed_ms)
        out = {'exposed_comm_ms': round(v[len(v) // 2], 3), 'segment_bytes': self.reducer.segment_bytes(),
               'segment_gather_bytes_per_rank': self.reducer.segment_gather_bytes()}
        if self._segment_marks:
            # median GPU time of each graph of the step (F: encoders forward, H: fusion + head forward / backward, T / V: encoder backward)
            seg = {}
            for j, name in enumerate(('F',) + tuple(self._order)):
                ts = sorted(m[j].elapsed_time(m[j + 1]) for m in self._segment_marks)
                seg[name] = round(ts[len(ts) // 2], 3)
            out['segment_ms'] = seg
        return out

    def __call__(self, batch: Optional[Dict[str, torch.Tensor]] = None) -> torch.Tensor:
        if batch is not None:
            for k, v in batch.items():
                self.static[k].copy_(v, non_blocking=True)
        if hasattr(self.opt, 'refresh_lr'):
            self.opt.refresh_lr()                        # a scheduler moved group['lr'] since the last replay: one small fill per changed step class
        timed = self.segmented and len(self._exposed_ms) < 512
        order = self._order
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(len(order) + 2)] if timed else None
        if timed:
            marks[0].record()
        self.g_main.replay()
        if self.segmented:
            red = self.reducer
            cur = torch.cuda.current_stream()
            xs = self._exchange_stream
            for i, name in enumerate(order):
                if timed:
                    marks[1 + i].record()
                self.graphs[name].replay()
                pack = self.graphs['pack' + name]
                if xs is None:
                    if pack is not None:
                        pack.replay()
                    red.reduce_segment(name)             # asynchronous: travels beside the next segment's graph
                else:
                    # the segment's whole exchange chain -- wire-format copies (fp32 -> bf16), all-reduce / all-gather, copies back into the
                    # fp32 arenas, scatter of the gathered embedding rows -- leaves the compute stream: ~0.5 ms of copy kernels per step
                    # (1.2 GB each way with bf16 buckets) run beside the next segment's graph instead of in front of / behind it
                    xs.wait_stream(cur)
                    with torch.cuda.stream(xs):
                        if pack is not None:
                            pack.replay()
                        red.reduce_segment(name)
                        red.wait_segment(name)
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()                                 # all compute of the step is enqueued: what follows is exposed exchange
            if xs is None:
                for name in order:
                    red.wait_segment(name)
            else:
                cur.wait_stream(xs)
            ev1.record()
            if timed:
                marks[len(order) + 1] = ev0
                self._exposed_ms.append((ev0, ev1))
                self._segment_marks.append(marks)
            self._sync_routed_counts()
            self.g_opt.replay()
        elif self.g_opt is not None:
            self.reducer.reduce_static()
            self._sync_routed_counts()
            self.g_opt.replay()
        return self.loss

    def _sync_routed_counts(self):
        """Dense MoE dispatch: an expert is updated when ANY rank routed a token to it (its reduced gradient is the same on every
        rank, so the decision must be too -- the warm-up steps included, they are real training steps)."""
        if getattr(self.reducer, 'single', True):
            return
        for m in self.model.modules():
            a = getattr(m, '_active', None)
            if a is not None and getattr(m, 'dense_dispatch', False):
                torch.distributed.all_reduce(a, group=self.reducer.group)
