"""Training / prediction loop for the coordinate networks (stands in for `pl.Trainer`).

The reference drives its models through `pytorch_lightning.Trainer.fit / .predict`
(reference launcher.py:156-165,179,213): per batch `training_step` -> `loss.backward()` ->
`Adam.step()`.  `Trainer` keeps that protocol.  For the model families of
BASELINE.json (SirenNet, BaseMLP, HashMLP with the fused tiny-MLP decoder) the step is
compiled into an explicit kernel chain, `FusedStep`, with no autograd graph:

    [hashgrid_fwd] -> linear_fwd x n -> mse_loss -> (linear_bwd_weight, linear_bwd_data) x n
    -> [hashgrid_bwd] -> [RCCL all-reduce of ONE flat gradient buffer] -> adam (ONE launch)

All activations live in workspaces allocated once per batch size; the encoder output and its
gradient use the feature-major layout so the hash kernels store/load coalesced.
Other models (e.g. the BatchNorm decoder) run `training_step` + autograd, op by op.
"""
import contextlib
import ctypes as C
import os
import time
from typing import Dict, List, Optional

import torch

from . import _lib, models, ops, optim, parallel
from .datamodules import BatchPipeline, DeviceLoader


class _Layer:
    def __init__(self, weight, bias, activation, w0):
        self.weight, self.bias, self.activation, self.w0 = weight, bias, activation, w0


def fusable_layers(model) -> Optional[tuple]:
    """(encoder or None, [_Layer, ...]) if the model is a plain chain of fused layers."""
    enc, layers = None, []
    if isinstance(model, models.ModulatedSirenNet):
        return None  # two interleaved stacks: runs training_step + autograd over the HIP ops
    if isinstance(model, models.SirenNet):
        for l in list(model.layers) + [model.last_layer]:
            if l._code is None:
                return None
            layers.append(_Layer(l.weight, l.bias, l._code[0], l._code[1]))
        return enc, layers
    if isinstance(model, models.HashMLP):
        enc = model.encoder
        for block in model.decoder:
            lin, norm, act, drop = block[0], block[1], block[2], block[3]
            if not isinstance(norm, models._Fused):
                return None
            if not isinstance(act, (models._Fused, torch.nn.Identity)):
                return None
            if drop.p != 0.0:
                return None
            layers.append(_Layer(lin.weight, lin.bias, lin.activation_code, lin.w0))
        return enc, layers
    if isinstance(model, models.BaseMLP) and isinstance(getattr(model, "layers", None),
                                                        torch.nn.Sequential):
        for m in model.layers:
            if isinstance(m, models.FusedLinear):
                layers.append(_Layer(m.weight, m.bias, m.activation_code, m.w0))
            elif not isinstance(m, models._Fused):
                return None
        return enc, layers
    return None


class FusedStep:
    """Explicit forward / backward kernel chain over preallocated workspaces."""

    def __init__(self, model, optimizer: optim.Adam, world: int = 1):
        plan = fusable_layers(model)
        if plan is None:
            raise ValueError("model is not a fusable chain")
        self.encoder, self.layers = plan
        if self.encoder is not None and getattr(self.encoder, "_too_fine", False):
            raise OverflowError("int too big to convert")  # what encoder.forward raises
        self.opt = optimizer
        self.flat = optimizer.flatten()
        self.world = world
        self.bwd_method = 0
        self.phase_events: Optional[Dict[str, list]] = None  # bench.py: per-phase HIP events
        self._ws = {}
        self.loss = torch.zeros(1, device=self.flat.param.device)
        self._grads = [(self.flat.grad_view(l.weight),
                        self.flat.grad_view(l.bias) if l.bias is not None else None)
                       for l in self.layers]
        self._table_grad = self.flat.grad_view(self.encoder.table) if self.encoder else None
        self.tiny = self._tiny_mlp_plan()
        self.use_tiny = self.tiny is not None
        self.chain = self._siren_chain_plan()
        self.use_chain = self.chain is not None
        # train_step folds the loss and the head's backward into the forward kernel (needs a
        # sine layer below the last one); forward() + backward() keep the separate kernels
        self.chain_loss = self.use_chain and len(self.layers) >= 3
        # one rank, no accumulation: the table's Adam update runs where its gradient is complete
        # (mri_hashgrid_backward_adam); the table part of flat.grad is then NOT produced.  Bit-identical
        # to the two launches; measured 0.574 -> 0.562 ms on config 4 but 0.451 -> 0.460 on config 2
        # and 0.869 -> 0.885 on config 5 (EXPERIMENTS.md Part II 4.2): off by default
        self.fuse_table_adam = False
        self._table_stepped = False
        # Data parallel: the table gradient is produced level by level, so its reduction is cut
        # into `grad_buckets` level groups; group g's all-reduce (RCCL, its own stream) runs
        # while group g+1's gradient is still being computed.  1 = one reduction at the end.
        # Default 1 for every world size: the bucketed form (asynchronous all-reduces of slices of one flat
        # buffer, per-group Adam behind each) has only ever met a one-rank RCCL communicator; it stays
        # opt-in (`grad_buckets = 4`, bench.py times it as its own leg) until a real multi-GPU run has
        # passed with it.
        self.grad_buckets = 1
        self._bucket_cache = None
        self.last_group_bytes = []  # bytes of each reduction of the last data-parallel step
        # "all_reduce": level-group all-reduces overlapped with the table-gradient kernels, every
        # rank steps the whole buffer.  "reduce_scatter": ONE reduce-scatter of the flat gradient,
        # each rank Adam-steps its 1/world shard (7/8 less optimiser traffic at 8 ranks), ONE
        # all-gather of the parameters -- same bytes on the links, but the all-gather sits between
        # Adam and the next forward pass instead of beside the gradient kernels (DESIGN.md
        # section 6); kept selectable so both can be timed on a real node.
        self._dp_mode = "all_reduce"
        self.rank = parallel.env_world()[0] if world > 1 else 0
        # Single launch of the table gradient: its counting stage depends on the coordinates
        # only, so it is queued on a side stream and overlaps the forward pass / decoder.
        self.overlap_count = True
        self._side = None
        self._counted = False
        self._bwd_ws = [None, None]  # scratch of the table gradient, see _hash_workspace: two of them,
        self._ws_index = 0           # so that the counting stage of batch k+1 can run during step k
        self._ahead = None           # what was counted ahead: dict(ptr, n, ws, event)
        self._count_event = None
        self._batch_event = None
        # Count batch k+1 during step k?  Pays when the decoder kernel starves the side stream: the
        # 128-wide bf16-pipe kernel holds every CU's whole register file (512 threads x 250 VGPRs), the
        # counting stage queued in its own step then ends ~30 us after it and the scatter waits
        # (config 4: 0.572 -> 0.561 ms).  The 64-wide kernel leaves room, there the extra work beside
        # the lookup only costs (config 2: 0.449 -> 0.458 ms).
        self.count_ahead = self.use_tiny and self.layers[0].weight.shape[0] == 128
        # the decoder kernel reports max |d_enc| per level, the scale of the table-gradient records
        # (two features per level: feature pair = level)
        self.decoder_absmax = bool(
            self.use_tiny and self.encoder is not None and self.encoder.n_features_per_level == 2
            and ops.tiny_mlp_dx_absmax_supported(self.layers[0].weight.shape[1],
                                                 self.layers[0].weight.shape[0]))
        self._absmax, self._absmax_index, self._absmax_clean = None, 0, [True, True]
        # Fused decoder, optional: cut the batch in two slices and run the encoder of the second
        # on its own stream beside the decoder kernel of the first (the decoder leaves the VALU
        # and the texture path idle, the encoder needs no LDS).  Measured at BASELINE config 4:
        # the half forward pass then costs 0.01 instead of 0.05 ms, but two decoder launches cost
        # 0.05 ms more than one (weights to LDS and gradient slabs twice) -- no net gain, so off.
        self.split_fraction = 0.0
        self._side2 = None
        self._split_rows = 0
        # Fused decoder, optional: the lookup runs BESIDE the decoder kernel (its own stream); the
        # decoder's workgroups wait per round for the slice of rows they are about to read
        # (ops.hashgrid_forward_signal / tiny_mlp_train_overlapped): one decoder launch, none of
        # the fixed costs of the two-slice form above.  Bit-identical results, but measured SLOWER
        # at BASELINE config 4 (0.87 against 0.66 ms per step): f32 MFMAs execute on the vector
        # ALUs, so beside the decoder's two MFMA waves per SIMD the lookup's ~150 vector
        # instructions per (coordinate, level) barely issue, and the 64 registers the decoder leaves
        # free hold 2 lookup waves per SIMD instead of 8 -- the lookup takes 0.49 ms instead of
        # 0.10 and the decoder waits for it (EXPERIMENTS.md Part II 4.7).  Off by default.
        self.overlap_forward = False
        self._ready = None       # uint64 slice counters: only ever grow
        self._ready_total = 0    # what a complete slice's counter holds after this step
        self._ready_key = None
        self._status = None
        self._overlapped = False

    @property
    def dp_mode(self) -> str:
        return self._dp_mode

    @dp_mode.setter
    def dp_mode(self, mode: str):
        """Refuse at configuration time what would otherwise fail inside the first step."""
        if mode not in ("all_reduce", "reduce_scatter"):
            raise ValueError(f"dp_mode {mode!r}: all_reduce or reduce_scatter")
        if mode == "reduce_scatter" and self.world > 1 and self.flat._grad_all.numel() % self.world:
            raise ValueError(f"reduce_scatter: the flat buffer ({self.flat._grad_all.numel()} floats, padded to "
                             f"256) does not divide over {self.world} ranks; use all_reduce")
        self._dp_mode = mode

    def _tiny_mlp_plan(self):
        """Parameters for the single-kernel tiny MLP (csrc/mlp_fused.hip) if the decoder is
        in -> H -> H -> 1 with ReLU hidden layers, a linear output and biases everywhere."""
        ls = self.layers
        if self.encoder is None or len(ls) != 3 or any(l.bias is None for l in ls):
            return None
        if [l.activation for l in ls] != [ops.ACT_RELU, ops.ACT_RELU, ops.ACT_IDENTITY]:
            return None
        h, k_in = ls[0].weight.shape
        if ls[1].weight.shape != (h, h) or ls[2].weight.shape != (1, h):
            return None
        if not ops.tiny_mlp_supported(k_in, h, 1):
            return None
        return dict(params=[(l.weight.data, l.bias.data) for l in ls], grads=self._grads)

    def _siren_chain_plan(self):
        """Arguments of the fused SIREN chain kernels (csrc/siren_chain.hip) if the model is
        dim_in -> 256 x n (sine) -> 1 (linear head) with biases everywhere."""
        ls = self.layers
        if self.encoder is not None or len(ls) < 2 or any(l.bias is None for l in ls):
            return None
        if any(l.activation != ops.ACT_SINE for l in ls[:-1]) or ls[-1].activation != ops.ACT_IDENTITY:
            return None
        hidden, dim_in = ls[0].weight.shape
        if any(l.weight.shape != (hidden, hidden) for l in ls[1:-1]) or ls[-1].weight.shape != (1, hidden):
            return None
        if len({l.w0 for l in ls[1:-1]}) > 1 or not ops.siren_supported(dim_in, hidden, len(ls) - 1, 1):
            return None
        return dict(weights=[l.weight.data for l in ls], biases=[l.bias.data for l in ls],
                    w0_first=ls[0].w0, w0=ls[1].w0 if len(ls) > 2 else ls[0].w0,
                    d_weights=[g[0] for g in self._grads], d_biases=[g[1] for g in self._grads])

    @contextlib.contextmanager
    def _phase(self, name: str):
        """Bracket a phase with HIP events on the launch stream when timing is enabled."""
        if self.phase_events is None:
            yield
            return
        start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
        yield
        stop.record()
        self.phase_events.setdefault(name, []).append((start, stop))

    def phase_ms(self) -> Dict[str, float]:
        """Mean milliseconds per step of every recorded phase (synchronises)."""
        torch.cuda.synchronize()
        return {k: sum(a.elapsed_time(b) for a, b in v) / len(v)
                for k, v in (self.phase_events or {}).items()}

    def _workspace(self, n: int, train: bool):
        key = (n, train)
        ws = self._ws.get(key)
        if ws is None:
            dev = self.flat.param.device
            new = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)  # noqa: E731
            # the chain kernels keep hidden activations on chip: inference needs the output only
            skip_hidden = self.use_chain and not train
            ws = dict(y=[None if skip_hidden and i < len(self.layers) - 1
                         else new(n, l.weight.shape[0]) for i, l in enumerate(self.layers)])
            if self.encoder is not None:
                ws["enc"] = new(self.encoder.output_dim, n)
            if train:
                ws["deriv"] = [new(n, l.weight.shape[0])
                               if l.activation in (ops.ACT_SINE, ops.ACT_GELU) else None
                               for l in self.layers]
                ws["dz"] = [new(n, l.weight.shape[0]) for l in self.layers]
                if self.encoder is not None:
                    ws["d_enc"] = new(self.encoder.output_dim, n)
            self._ws = {k: v for k, v in self._ws.items() if k[1] != train}  # keep one size
            self._ws[key] = ws
        return ws

    def forward(self, coords: torch.Tensor, train: bool = False):
        n = coords.shape[0]
        ws = self._workspace(n, train)
        x, feature_major = coords, False
        self._overlapped = False
        if self.encoder is not None and train and self.use_tiny and self._overlap_plan(n):
            # lookup on its own stream, queued BEFORE the decoder (which backward() queues on the
            # main stream and which waits for the slices through the ready counters)
            self._side2.wait_stream(torch.cuda.current_stream())  # coordinates, tables, buffers
            with torch.cuda.stream(self._side2):
                with self._phase("hashgrid_fwd"):
                    ops.hashgrid_forward_signal(self.encoder.desc, coords, self.encoder.table.data,
                                                ws["enc"], self._ready_key[1], self._ready)
            self._ready_total += self._ready_expected
            self._overlapped = True
            return ws["enc"], ws
        if self.encoder is not None:
            h = 0
            if train and self.use_tiny and self.split_fraction > 0 and n >= 4096:
                # the second slice is encoded on its own stream beside the first slice's decoder
                # kernel, which backward() queues
                h = max(32, int(n * self.split_fraction) // 32 * 32)
            self._split_rows = h  # rows [h, n) are encoded by backward(), see there
            with self._phase("hashgrid_fwd"):
                x = ops.hashgrid_forward(self.encoder.desc, coords[:h] if h else coords,
                                         self.encoder.table.data, out=ws["enc"],
                                         feature_major=True)
            feature_major = True
        if self.use_tiny and not train:
            with self._phase("mlp_fwd"):
                return ops.tiny_mlp_forward(x, self.tiny["params"], y=ws["y"][-1]), ws
        if self.use_tiny:
            return x, ws  # the training kernel runs forward and backward together
        if self.use_chain:  # every layer of a row tile in one kernel, activations stay in LDS
            c, n_sine = self.chain, len(self.layers) - 1
            with self._phase("mlp_fwd"):
                ops.siren_forward(x, c["weights"], c["biases"], c["w0_first"], c["w0"],
                                  act=ws["y"][:n_sine] if train else None,
                                  deriv=ws["deriv"][:n_sine] if train else None, y=ws["y"][-1])
            return ws["y"][-1], ws
        with self._phase("mlp_fwd"):
            for i, l in enumerate(self.layers):
                deriv = ws["deriv"][i] if train else None
                x = ops.linear_forward(x, l.weight.data,
                                       None if l.bias is None else l.bias.data, l.activation,
                                       l.w0, out=ws["y"][i], deriv=deriv,
                                       x_feature_major=feature_major)
                feature_major = False
        return x, ws

    def _hash_workspace(self, n: int):
        """Scratch of the table-gradient kernels, owned by this FusedStep (two models in one
        process never share it).  It is used on the main AND the side stream; when a larger batch
        needs a larger buffer, both streams are drained before the old one goes back to the
        allocator."""
        return self._hash_workspace_at(self._ws_index, n)

    def _hash_workspace_at(self, index: int, n: int):
        need = ops.backward_workspace_bytes(self.encoder.desc, n)
        ws = self._bwd_ws[index]
        if ws is None or ws.numel() * 8 < need:
            if ws is not None:
                torch.cuda.current_stream().synchronize()
                if self._side is not None:
                    self._side.synchronize()
            ws = torch.empty((need + 7) // 8, dtype=torch.int64, device=self.flat.param.device)
            self._bwd_ws[index] = ws
        return ws

    def forget_ahead(self):
        """Drop what was counted for a batch that will not be the next one (a new loader, say)."""
        self._ahead = None

    def _await_count(self):
        """The main stream waits for the counting stage of THIS batch: the event recorded behind it when
        it ran a step ahead, else everything queued on the side stream."""
        if self._count_event is not None:
            torch.cuda.current_stream().wait_event(self._count_event)
            self._count_event = None
        else:
            torch.cuda.current_stream().wait_stream(self._side)

    def _overlap_plan(self, n: int) -> bool:
        """Prepare the slice counters of the side-by-side lookup for a batch of n rows."""
        if not self.overlap_forward or self.split_fraction > 0:
            return False
        k_in, hidden = self.layers[0].weight.shape[1], self.layers[0].weight.shape[0]
        rows = ops.tiny_mlp_round_rows(k_in, hidden, n)
        expected = ops.hashgrid_signal_blocks(self.encoder.desc, rows) if rows > 0 else -1
        if rows <= 0 or expected <= 0:
            self.overlap_forward = False  # other decoder widths / grids: lookup first, as before
            return False
        if self._ready_key != (n, rows):
            dev = self.flat.param.device
            if self._ready is not None:
                torch.cuda.synchronize(dev)  # nobody may still be counting on the old layout
            self._ready = torch.zeros(-(-n // rows), dtype=torch.int64, device=dev)
            self._status = torch.zeros(1, dtype=torch.int32, device=dev)
            self._ready_total, self._ready_key, self._ready_expected = 0, (n, rows), expected
            if self._side2 is None:
                self._side2 = torch.cuda.Stream(device=dev)
        return True

    def check_status(self):
        """Raise if a decoder workgroup ever gave up waiting for the lookup (synchronises)."""
        if self._status is not None and int(self._status.item()) != 0:
            raise RuntimeError("fused step: the decoder kernel timed out waiting for the hash-grid "
                               "lookup running beside it (results of that step are invalid)")

    def _level_buckets(self):
        """[(level mask, flat-gradient slice)] in execution order for the current bucket count
        (see level_group_ranges)."""
        enc = self.encoder
        if self._bucket_cache is not None and self._bucket_cache[0] == self.grad_buckets:
            return self._bucket_cache[1]
        table_off = self.flat.offsets[next(k for k, q in enumerate(self.flat.params)
                                           if q is enc.table)]
        out = [(mask, self.flat.grad[lo:hi])
               for mask, lo, hi in gradient_group_slices(list(enc.sizes),
                                                         enc.n_features_per_level, table_off,
                                                         self.flat.numel, self.grad_buckets)
               if mask is not None]
        self._bucket_cache = (self.grad_buckets, out)
        return out

    def _take_absmax(self):
        """A zeroed buffer for the decoder's max |d_enc| per level, or None if this decoder / encoder
        pair does not use one.  Two buffers alternate: the one for the NEXT step is zeroed on the side
        stream beside this step (train_step: queue_side), otherwise here, on the main stream."""
        if not self.decoder_absmax:
            return None
        w1 = self.layers[0].weight  # (the library option that selects the decoder kernel may have changed)
        if not ops.tiny_mlp_dx_absmax_supported(w1.shape[1], w1.shape[0]):
            return None
        if self._absmax is None:
            self._absmax = torch.zeros(2, 32, device=self.flat.param.device)
            self._absmax_clean = [True, True]
        i = self._absmax_index = 1 - self._absmax_index
        if not self._absmax_clean[i]:
            self._absmax[i].zero_()
        self._absmax_clean[i] = False
        return self._absmax[i]

    def _hash_backward(self, coords, d_enc, overwrite=False, reduce=True, absmax=None):
        """Table gradient; with several ranks, reduce each finished level group right away
        (`reduce=False`: a micro-batch of an accumulation group that does not step)."""
        enc = self.encoder
        if self._counted:
            self._await_count()
        counted, self._counted = self._counted, False
        ws = self._hash_workspace(coords.shape[0]) if self.bwd_method != 1 else None
        if self.grad_buckets <= 1 or not reduce or self.dp_mode == "reduce_scatter":
            ops.hashgrid_backward(enc.desc, coords, d_enc, self._table_grad, feature_major=True,
                                  method=self.bwd_method, prepared=counted, overwrite=overwrite,
                                  ws=ws, level_absmax=absmax)
            return []
        if not counted and self.bwd_method != 1:  # one count for all the groups
            ops.hashgrid_backward_prepare(enc.desc, coords, self.bwd_method, ws=ws)
            counted = True
        pending = []
        for mask, grad_slice in self._level_buckets():
            ops.hashgrid_backward(enc.desc, coords, d_enc, self._table_grad, feature_major=True,
                                  method=self.bwd_method, prepared=counted, overwrite=overwrite,
                                  level_mask=mask, ws=ws, level_absmax=absmax)
            pending.append(self._reduce_async(grad_slice))
        return pending

    def _table_range(self):
        f = self.encoder.n_features_per_level
        t0 = self.flat.offsets[next(k for k, q in enumerate(self.flat.params)
                                    if q is self.encoder.table)]
        return t0, t0 + self.encoder.table.shape[0] * f

    def _hash_backward_adam(self, coords, d_enc) -> bool:
        """Table gradient + the table's Adam step in one pass; False if this grid / configuration
        cannot take it (the caller then runs the two-launch form)."""
        if not (self.fuse_table_adam and self.world == 1 and self.bwd_method != 1):
            return False
        enc = self.encoder
        if self._counted:
            self._await_count()
        counted = self._counted
        t0, t1 = self._table_range()
        o = self.opt
        ok = ops.hashgrid_backward_adam(
            enc.desc, coords, d_enc, self.flat.param[t0:t1], self.flat.exp_avg[t0:t1],
            self.flat.exp_avg_sq[t0:t1], o.param_groups[0]["lr"], o.betas[0], o.betas[1], o.eps,
            o.step_count + 1, o.grad_scale, feature_major=True, method=self.bwd_method,
            prepared=counted, ws=self._hash_workspace(coords.shape[0]))
        if not ok:  # nothing was done: the ordinary path computes the gradient and opens the step
            return False
        o.begin_step()  # step number step_count + 1 is open; train_step steps the rest of the buffer
        self._counted = False
        self._table_stepped = True
        return True

    def _reduce_async(self, grad_slice):
        """(handle, lo, hi): a started reduction of flat.grad[lo:hi]."""
        lo = grad_slice.storage_offset() - self.flat.grad.storage_offset()
        return parallel.all_reduce_async(grad_slice), lo, lo + grad_slice.numel()

    def _reduce_decoder_grads(self):
        """Start the reduction of everything in the flat gradient buffer that is not the hash
        table (the decoder's weights: complete as soon as its backward kernels are queued), so
        that it runs beside the table-gradient kernels instead of behind the table's groups."""
        if self.world <= 1 or self.grad_buckets <= 1 or self.encoder is None \
                or self.dp_mode == "reduce_scatter":
            return []
        f = self.encoder.n_features_per_level
        t0 = self.flat.offsets[next(k for k, q in enumerate(self.flat.params)
                                    if q is self.encoder.table)]
        t1 = t0 + self.encoder.table.shape[0] * f
        out = []
        if t0 > 0:
            out.append(self._reduce_async(self.flat.grad[:t0]))
        if t1 < self.flat.numel:
            out.append(self._reduce_async(self.flat.grad[t1:]))
        return out

    def _deriv_of(self, i, ws):
        mode = ops.deriv_mode_for(self.layers[i].activation)
        if mode == ops.DERIV_MUL:
            return mode, ws["deriv"][i]
        if mode == ops.DERIV_RELU_MASK:
            return mode, ws["y"][i]
        return mode, None

    def backward(self, coords, target, ws, first=True, step=True, divisor=1.0):
        """Gradients of mean squared error into the flat gradient buffer.  `first`: the buffer
        is started afresh (zeroed / overwritten), else added to; `step`: this call completes the
        gradient (data parallel: start its reductions); gradients are scaled by
        1 / (world * divisor)."""
        div = float(self.world) * float(divisor)
        if self.use_tiny:
            # every gradient (tables, decoder) and the loss are OVERWRITTEN by the two kernels
            # below: no zeroing pass over the flat gradient buffer
            with self._phase("mlp_fused"):
                h, n = self._split_rows, coords.shape[0]
                if not first:
                    self.loss.zero_()  # the kernels add to it: keep it this batch's loss
                absmax = None
                if self._overlapped:
                    ops.tiny_mlp_train_overlapped(ws["enc"], target, self.tiny["params"],
                                                  self.tiny["grads"], self.loss, ws["d_enc"],
                                                  self._ready, self._ready_total, self._status,
                                                  grad_divisor=div, overwrite=first)
                elif h:
                    if self._side2 is None:
                        self._side2 = torch.cuda.Stream(device=coords.device)
                    self._side2.wait_stream(torch.cuda.current_stream())  # starts with slice 1's decoder
                    with torch.cuda.stream(self._side2):
                        ops.hashgrid_forward(self.encoder.desc, coords[h:],
                                             self.encoder.table.data, out=ws["enc"],
                                             feature_major=True, row_offset=h)
                    ops.tiny_mlp_train_slice(ws["enc"], target, 0, h, self.tiny["params"],
                                             self.tiny["grads"], self.loss, ws["d_enc"],
                                             grad_divisor=div, overwrite=first)
                    torch.cuda.current_stream().wait_stream(self._side2)  # second slice encoded
                    ops.tiny_mlp_train_slice(ws["enc"], target, h, n - h, self.tiny["params"],
                                             self.tiny["grads"], self.loss, ws["d_enc"],
                                             grad_divisor=div, overwrite=False)
                else:
                    # a whole batch in one decoder call: the kernel reports max |d_enc| per level
                    # on the way (the scale of the table-gradient records), no pass over d_enc
                    absmax = self._take_absmax() if (first and step) else None
                    ops.tiny_mlp_train(ws["enc"], target, self.tiny["params"],
                                       self.tiny["grads"], self.loss, d_x=ws["d_enc"],
                                       grad_divisor=div, overwrite=first, dx_absmax=absmax)
            started = self._reduce_decoder_grads() if step else []
            with self._phase("hashgrid_bwd"):
                if first and step and self._hash_backward_adam(coords, ws["d_enc"]):
                    self._pending = []
                else:
                    self._pending = started + self._hash_backward(coords, ws["d_enc"],
                                                                  overwrite=first, reduce=step,
                                                                  absmax=absmax)
            return
        with self._phase("zero_grad"):
            if first:
                self.flat.grad.zero_()
            self.loss.zero_()
        last = len(self.layers) - 1
        pred = ws["y"][last]
        dz = ws["dz"][last]
        with self._phase("loss"):
            ops.mse_loss(pred, target, self.loss, dz, grad_divisor=div)
            mode, g = self._deriv_of(last, ws)
            ops.apply_deriv(dz, mode, g)
        if self.use_chain:
            c, n_sine = self.chain, last
            with self._phase("mlp_bwd"):
                ops.siren_backward(coords, dz, c["weights"], ws["y"][:n_sine], ws["deriv"][:n_sine],
                                   [None] + ws["dz"][1:n_sine], c["d_weights"], c["d_biases"])
            return
        with self._phase("mlp_bwd"):
            for i in range(last, -1, -1):
                l = self.layers[i]
                gw, gb = self._grads[i]
                first_on_encoder = i == 0 and self.encoder is not None
                x = ws["enc"] if first_on_encoder else (coords if i == 0 else ws["y"][i - 1])
                ops.linear_backward_weight(dz, x, gw, gb, x_feature_major=first_on_encoder)
                if i > 0:
                    mode, g = self._deriv_of(i - 1, ws)
                    dz = ops.linear_backward_data(dz, l.weight.data, mode, g,
                                                  dx=ws["dz"][i - 1])
                elif self.encoder is not None:
                    ops.linear_backward_data(dz, l.weight.data, ops.DERIV_NONE, None,
                                             dx=ws["d_enc"], dx_feature_major=True)
        if self.encoder is not None:
            started = self._reduce_decoder_grads() if step else []
            with self._phase("hashgrid_bwd"):
                self._pending = started + self._hash_backward(coords, ws["d_enc"], reduce=step)

    def _chain_loss_pass(self, coords, target, first, divisor):
        """forward + loss + backward of the SIREN chain in three launches: the forward kernel also
        takes the loss and runs the head's backward while the last sine layer's output is still in
        registers (it is never written), the backward kernel starts from that."""
        ws = self._workspace(coords.shape[0], True)
        c, n_sine = self.chain, len(self.layers) - 1
        with self._phase("zero_grad"):
            if first:
                self.flat.grad.zero_()
            self.loss.zero_()
        act, deriv = ws["y"][:n_sine], ws["deriv"][:n_sine]
        with self._phase("mlp_fwd"):
            ops.siren_forward_loss(coords, target, c["weights"], c["biases"], c["w0_first"], c["w0"],
                                   act[:-1] + [None], deriv[:-1] + [None], ws["dz"][n_sine - 1],
                                   ws["y"][-1], c["d_weights"][-1], c["d_biases"][-1],
                                   c["d_biases"][n_sine - 1], self.loss,
                                   grad_divisor=float(self.world) * float(divisor))
        with self._phase("mlp_bwd"):
            ops.siren_backward(coords, None, c["weights"], act[:-1] + [None], deriv[:-1] + [None],
                               [None] + ws["dz"][1:n_sine], c["d_weights"], c["d_biases"],
                               head_done=True)

    def train_step(self, coords, target, side_work=None, first=True, step=True,
                   divisor=1.0, late_work=None) -> torch.Tensor:
        """One batch: forward, loss, backward and -- with `step` -- gradient reduction and Adam;
        returns the (device) loss scalar of this rank's batch.  Gradient accumulation over k
        batches: first=True on the first one, step=True on the last, divisor=k on all.
        `side_work()` (e.g. BatchPipeline.produce_next) is queued where it overlaps the step:
        on the side stream behind the counting stage when there is one, else after Adam.
        Contract: if side_work() returns a CUDA tensor, that tensor IS the `coords` of the next
        train_step call and is not modified in between -- with `count_ahead` its table-gradient
        records are counted during this step (forget_ahead() drops such a count).
        `late_work()` (e.g. BatchPipeline.produce_late: the production of a whole GROUP of later batches)
        is queued on the same stream behind that count, so that nothing the next step waits for sits
        behind it."""
        if self._batch_event is not None:  # this batch was produced on the side stream during the last step
            torch.cuda.current_stream().wait_event(self._batch_event)
            self._batch_event = None
        if self.encoder is not None and self.overlap_count and self.bwd_method != 1:
            if self._side is None:
                # (stream priorities, measured in round 4: this runtime offers 0 and -1 only; the main stream on -1
                # delays the side work until it lands beside the table gradient: lookup -2 us, table gradient
                # +10 us, step unchanged.  Both streams stay at the default priority.)
                self._side = torch.cuda.Stream(device=coords.device)
            # after the coordinates exist and after the previous step's backward released its
            # workspace and batch buffer (all earlier work of the current stream)
            self._side.wait_stream(torch.cuda.current_stream())
            ahead, self._ahead = self._ahead, None
            if ahead is not None and ahead["ptr"] == coords.data_ptr() and ahead["n"] == coords.shape[0]:
                # this batch was counted during the previous step (below), into the other workspace
                self._ws_index, self._count_event = ahead["ws"], ahead["event"]
            else:
                self._count_event = None
                ops.hashgrid_backward_prepare(self.encoder.desc, coords, self.bwd_method, self._side,
                                              ws=self._hash_workspace(coords.shape[0]))
            self._counted = True
            if side_work is not None:
                work, side_work = side_work, None

                def queue_side():
                    with torch.cuda.stream(self._side):
                        nxt = work()
                        if self._absmax is not None:
                            # the buffer of the step BEFORE this one (free since the wait_stream above)
                            # is the next step's: zero it here, behind _batch_event (see _take_absmax)
                            j = self._absmax_index
                            self._absmax[j].zero_()
                            self._absmax_clean[j] = True
                    # the next step's forward pass reads what side_work produced: it waits for this
                    # event (the wait for the counting stage no longer covers it once that runs ahead)
                    self._batch_event = torch.cuda.Event()
                    self._batch_event.record(self._side)
                    if self.count_ahead and torch.is_tensor(nxt) and nxt.is_cuda:
                        # `side_work` produced the NEXT batch and returned its coordinates: count it
                        # now, into the workspace this step does not use.  The side stream is starved
                        # while the decoder kernel holds every CU's registers, so a count queued in its
                        # own step finished ~30 us after the decoder and the scatter waited for it; a
                        # step ahead it has a whole step to finish
                        other = 1 - self._ws_index
                        ops.hashgrid_backward_prepare(self.encoder.desc, nxt, self.bwd_method,
                                                      self._side,
                                                      ws=self._hash_workspace_at(other, nxt.shape[0]))
                        ev = torch.cuda.Event()
                        ev.record(self._side)
                        self._ahead = dict(ptr=nxt.data_ptr(), n=nxt.shape[0], ws=other, event=ev)
                    if late is not None:
                        with torch.cuda.stream(self._side):
                            late()
                        # (a later step's wait for ITS batch event, recorded on this stream, covers it)

                late, late_work = late_work, None
                queue_side()
        if self.use_chain and self.chain_loss:
            self._pending = []
            self._chain_loss_pass(coords, target, first, divisor)
        else:
            _, ws = self.forward(coords, train=True)
            self._pending = []
            self.backward(coords, target, ws, first, step, divisor)
        if not step:
            if side_work is not None:
                side_work()
            if late_work is not None:
                late_work()
            return self.loss
        if self.world > 1 and self.dp_mode == "reduce_scatter":
            with self._phase("all_reduce"):
                all_grads = self.flat._grad_all
                self.last_group_bytes = [all_grads.numel() * 4, all_grads.numel() * 4]  # scatter, gather
                with self._phase("reduce_wait_0"):
                    parallel.reduce_scatter_sum(all_grads, self.rank, self.world)
                lo, hi = parallel.shard_range(all_grads.numel(), self.rank, self.world)
                self.opt.begin_step()
                self.opt.step_shard(lo, hi)
                self.opt.sharded = True  # the moments of the other shards are stale from here on
                with self._phase("reduce_wait_1"):
                    parallel.all_gather_shards(self.flat._param_all, self.rank, self.world)
        elif self.world > 1 and self._pending:
            # decoder and table level groups are already in flight, in this order; each one is
            # stepped as soon as its sum has landed, beside the reductions still running (Adam is
            # elementwise: the pieces give bit for bit what one launch over the buffer gives)
            if not covers_exactly_once(self._pending, self.flat.numel):
                raise RuntimeError("gradient groups do not cover the flat buffer exactly once")
            self.last_group_bytes = [(hi - lo) * 4 for _, lo, hi in self._pending]
            with self._phase("all_reduce"):
                self.opt.begin_step()
                for i, (handle, lo, hi) in enumerate(self._pending):
                    with self._phase(f"reduce_wait_{i}"):  # how long the compute stream stood still for it
                        parallel.wait_all([handle])
                    self.opt.step_range(lo, hi)
        elif self._table_stepped:  # the table was stepped with its gradient: the rest of the buffer
            self._table_stepped = False
            t0, t1 = self._table_range()
            with self._phase("adam"):
                if t0 > 0:
                    self.opt.step_range(0, t0)
                if t1 < self.flat.numel:
                    self.opt.step_range(t1, self.flat.numel)
        else:
            if self.world > 1:
                self.last_group_bytes = [self.flat.grad.numel() * 4]
                with self._phase("all_reduce"), self._phase("reduce_wait_0"):
                    parallel.all_reduce_sum(self.flat.grad)
            with self._phase("adam"):
                self.opt.step()
        if side_work is not None:
            side_work()
        if late_work is not None:
            late_work()
        return self.loss


class SteadyLoop:
    """The fused hash-grid + tiny-MLP training step queued with little host work per step.

    Why: a step is ~12 kernel launches, a few memsets and a stream fork / join; queued op by op from Python
    through ctypes and torch's stream / event objects that is 0.2-0.55 ms of host time per step depending on the
    box (tools/host_profile.py: 57 % of it interpreter and wrapper overhead) against 0.5 ms of GPU time -- on
    a slow host the loop is HOST-bound (0.72 ms per step measured).  Here a step is ONE call of the library's
    `mri_fused_step`, which composes the same entry points in C (csrc/fused_step.hip).  (Round 3 also built the
    loop as hipGraph replays: correct, and on this runtime a replay cost the host as much as the eager step --
    EXPERIMENTS.md; removed in round 4.)
    It is FusedStep.train_step's steady state with count_ahead:
        side stream: produce batch k+1 (sample + gather) -> zero the next absmax buffer -> count batch k+1's
                     table-gradient records;    main: lookup -> decoder -> table gradient -> Adam;    join.
    The batch pipeline's two buffers, the two record workspaces and the two absmax buffers alternate with
    the step's parity.  Same launches on the same data in the same order as the eager step: parameters are
    bit-identical (tests/test_gpu_round3.py::test_steady_loop_equals_the_eager_loop).

    Eager steps (bench.py's --launch eager, evaluation passes) can be mixed in: the Python
    state of FusedStep / BatchPipeline / Adam is advanced as the eager step would have left it.  Two limits:
    (1) the native argument blocks hold raw device addresses of FusedStep's buffers; an eager step with ANOTHER
    batch size makes FusedStep reallocate them -- step_once() detects that and raises (capture() again);
    (2) after a natively queued step `flat.grad` does NOT hold the table gradient of the levels whose sums meet
    in the int64 workspace (the dense levels, bins cut over entry ranges): their conversion is folded into the
    Adam kernel (csrc/fused_step.hip), which consumes them straight from the workspace.  Readers of `.grad`
    (gradient clipping, logging) must take an eager step, whose table gradient is complete in `flat.grad`."""

    @staticmethod
    def unsupported(step: "FusedStep", pipe: BatchPipeline) -> Optional[str]:
        ld = pipe.loader
        if step.world != 1 and (step.dp_mode != "all_reduce" or step.grad_buckets > 1):
            return "several ranks: the plain all-reduce form only (one reduction of the flat gradient)"
        if not (step.use_tiny and step.encoder is not None):
            return "the fused hash-grid + tiny-MLP step only"
        if step.overlap_forward or step.split_fraction > 0 or step.fuse_table_adam or step.bwd_method == 1:
            return "an optional step form is selected"
        if not step.overlap_count:
            return "needs the counting stage on the side stream"
        if pipe.group != 1 or not ld.shuffle:
            return "needs a shuffled loader and BatchPipeline(group=1)"
        if ld.steps is None and not ld.drop_last and (ld.hi - ld.lo) % ld.batch_size:
            return "every batch must be full (drop_last, a fixed step count, or a divisible range)"
        if ld.hi - ld.lo < ld.batch_size:
            return "range shorter than a batch"
        return None

    def __init__(self, step: "FusedStep", pipe: BatchPipeline, mode: str = "native"):
        why = self.unsupported(step, pipe)
        if why:
            raise ValueError("SteadyLoop: " + why)
        if mode != "native":
            raise ValueError("SteadyLoop: one form, mode='native' (the hipGraph-replay form was removed in round 4)")
        self.step, self.pipe, self.mode = step, pipe, mode
        self._wmap = self._amap = None
        self._after_eager = True
        self._join_pending = False
        self._args = [None, None]
        self._pins = [None, None]
        self._ev_fork, self._ev_join = torch.cuda.Event(), torch.cuda.Event()

    # -- the buffers of a step of parity p (batch k in pipeline buffer p) --------------------------------
    def _buffers(self, p: int):
        st, pipe = self.step, self.pipe
        n = pipe.loader.batch_size
        q = 1 - p
        _, coords_p, target_p = (t[:n] for t in pipe.slots[p])
        idx_q, coords_q, target_q = (t[:n] for t in pipe.slots[q])
        ws_p, ws_q = st._bwd_ws[self._wmap[p]], st._bwd_ws[self._wmap[q]]
        am_p = st._absmax[self._amap[p]] if st._absmax is not None else None
        am_q = st._absmax[self._amap[q]] if st._absmax is not None else None
        return n, coords_p, target_p, idx_q, coords_q, target_q, ws_p, ws_q, am_p, am_q

    def _native_args(self, p: int) -> "_lib.FusedStepArgs":
        """Native mode: the argument block of parity p; everything but the per-step scalars is fixed."""
        st = self.step
        ld, ds, enc = self.pipe.loader, self.pipe.loader.ds, st.encoder
        n, coords_p, target_p, idx_q, coords_q, target_q, ws_p, ws_q, am_p, am_q = self._buffers(p)
        w = st._workspace(n, True)
        ptr = lambda t: None if t is None else t.data_ptr()  # noqa: E731
        a = _lib.FusedStepArgs()
        a.grid = C.pointer(enc.desc)
        a.table = ptr(enc.table.data)
        (w1, b1), (w2, b2), (w3, b3) = st.tiny["params"]
        (g1, gb1), (g2, gb2), (g3, gb3) = st.tiny["grads"]
        a.w1, a.b1, a.w2, a.b2, a.w3, a.b3 = (ptr(t) for t in (w1, b1, w2, b2, w3, b3))
        a.d_w1, a.d_b1, a.d_w2, a.d_b2, a.d_w3, a.d_b3 = (ptr(t) for t in (g1, gb1, g2, gb2, g3, gb3))
        a.d_table, a.loss = ptr(st._table_grad), ptr(st.loss)
        a.hidden, a.bwd_method, a.counted = int(w1.shape[0]), int(st.bwd_method), 1
        a.coords, a.target, a.n = ptr(coords_p), ptr(target_p), n
        a.enc, a.d_enc = ptr(w["enc"]), ptr(w["d_enc"])
        tiny_ws = ops._tiny_workspace(w1.shape[1], w1.shape[0], n, coords_p.device)
        a.tiny_ws, a.tiny_ws_bytes = ptr(tiny_ws), tiny_ws.numel() * 4
        a.bwd_ws, a.bwd_ws_bytes = ptr(ws_p), ws_p.numel() * 8
        if not st.count_ahead:  # the 64-wide decoder: every batch is counted inside its own step
            a.counted = 0
        a.absmax = ptr(am_p)
        f, o = st.flat, st.opt
        a.param, a.grad, a.exp_avg, a.exp_avg_sq = ptr(f.param), ptr(f.grad), ptr(f.exp_avg), ptr(f.exp_avg_sq)
        a.n_params = f.numel if st.world == 1 else 0  # several ranks: reduce, then step (step_once)
        a.lr, a.beta1, a.beta2, a.eps = o.param_groups[0]["lr"], o.betas[0], o.betas[1], o.eps
        a.grad_scale, a.grad_divisor = o.grad_scale, float(st.world)
        a.next_idx, a.next_coords, a.next_target, a.next_n = ptr(idx_q), ptr(coords_q), ptr(target_q), n
        a.next_bwd_ws, a.next_bwd_ws_bytes, a.next_absmax = ptr(ws_q), ws_q.numel() * 8, ptr(am_q)
        if not st.count_ahead:
            a.next_bwd_ws, a.next_bwd_ws_bytes = None, 0
        a.lo, a.hi, a.dim = ld.lo, ld.hi, ds.dim_in
        for d in range(ds.dim_in):
            a.shape[d], a.axis_offset[d] = int(ds.shape[d]), int(ds.axis_offset[d])
        a.axes, a.volume = ptr(ds.axes), ptr(ds.pixels)
        a.stream_side = st._side.cuda_stream
        a.ev_fork, a.ev_join = self._ev_fork.cuda_event, self._ev_join.cuda_event
        # The struct holds RAW pointers: every tensor behind one is pinned here (kept alive) with the address it
        # had, and step_once() refuses to queue through a block whose tensors have moved -- FusedStep evicts
        # its per-size workspaces (`_workspace`: one size kept) and regrows `_bwd_ws` when an eager step with
        # another batch size runs in between, which would otherwise leave the block writing through stale addresses.
        pins = dict(enc=w["enc"], d_enc=w["d_enc"], tiny_ws=tiny_ws, ws_p=ws_p, ws_q=ws_q, table_grad=st._table_grad,
                    param=f.param, grad=f.grad, exp_avg=f.exp_avg, exp_avg_sq=f.exp_avg_sq, coords_p=coords_p,
                    coords_q=coords_q, idx_q=idx_q)
        if am_p is not None:
            pins.update(absmax_p=am_p, absmax_q=am_q)
        self._pins[p] = [(name, t, t.data_ptr()) for name, t in pins.items()]
        return a

    def _check_pins(self, p: int):
        st = self.step
        n = self.pipe.loader.batch_size
        live = dict(enc=st._ws.get((n, True), {}).get("enc"), d_enc=st._ws.get((n, True), {}).get("d_enc"),
                    ws_p=st._bwd_ws[self._wmap[p]], ws_q=st._bwd_ws[self._wmap[1 - p]], table_grad=st._table_grad,
                    param=st.flat.param, grad=st.flat.grad)
        for name, t, addr in self._pins[p]:
            now = live.get(name, t)
            if now is None or now.data_ptr() != addr:
                raise RuntimeError(f"SteadyLoop: the {name} buffer of FusedStep moved since capture() (an eager step "
                                   "with another batch size ran in between?): call capture() again")

    def capture(self, warm_steps: int = 4):
        """`warm_steps` eager steps (at least two before the first capture: they allocate every workspace and
        leave the next batch produced and counted), then the per-parity argument blocks."""
        st, pipe = self.step, self.pipe
        for _ in range(warm_steps):  # (0: the caller's own eager steps have established the steady state)
            self.eager_step()
        torch.cuda.synchronize()
        a = st._ahead
        coords, _ = pipe.current()
        p0 = pipe.k % 2
        if not st.count_ahead:
            self._wmap = {0: st._ws_index, 1: st._ws_index}
        elif a is None or a["ptr"] != coords.data_ptr():
            raise RuntimeError("SteadyLoop: the eager steps did not leave the next batch counted")
        else:
            self._wmap = {p0: a["ws"], 1 - p0: 1 - a["ws"]}
        nxt = 1 - st._absmax_index
        self._amap = {p0: nxt, 1 - p0: 1 - nxt}
        if st._absmax is not None and not st._absmax_clean[nxt]:
            st._absmax[nxt].zero_()
            st._absmax_clean[nxt] = True
        st._batch_event = None
        torch.cuda.synchronize()
        self._ev_fork.record()  # materialises the handles
        self._ev_join.record()
        torch.cuda.synchronize()
        self._pins = [None, None]
        self._args = [self._native_args(0), self._native_args(1)]
        self._after_eager, self._join_pending = True, False
        return self

    def eager_step(self, **kw):
        """One step through FusedStep.train_step (same state protocol), e.g. with phase events enabled."""
        if self._join_pending:  # the previous native step's side work (this batch, its count)
            torch.cuda.current_stream().wait_event(self._ev_join)
            self._join_pending = False
        coords, target = self.pipe.current()
        loss = self.step.train_step(coords, target, self.pipe.produce_next, late_work=self.pipe.produce_late, **kw)
        self.pipe.advance()
        self._after_eager = True
        return loss

    PHASES = ("hashgrid_fwd", "mlp_fused", "hashgrid_bwd", "adam")

    def reserve_samples(self, n: int):
        """Create and materialise the timing events of `n` sampled steps NOW, outside the timed region: creating
        a timing event and recording it for the first time inside a deep queue stalled the host for milliseconds
        on some boxes (bench.py's 20-step form with five sampled steps: 2-4 ms per step instead of 0.52)."""
        pool = getattr(self, "_event_pool", [])
        while len(pool) < n:
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
            for ev in evs:
                ev.record()  # materialises the handle (re-recorded inside the library call)
            pool.append(evs)
        self._event_pool = pool

    def step_once(self, sample: bool = False):
        """Queue the current step; returns the (device) loss scalar.  `sample`: bracket the four phases with
        timing events INSIDE the library call and file them under FusedStep.phase_events (an eager sample
        step between queued ones stalls the queue for over a millisecond on this runtime)."""
        st, pipe = self.step, self.pipe
        k = pipe.k
        p = k % 2
        ld, opt = pipe.loader, st.opt
        if self._after_eager:  # the eager step's side-stream work (next batch, its count) must be done
            torch.cuda.current_stream().wait_stream(st._side)
            st._batch_event = None
            a = st._ahead
            if (st.count_ahead and (a is None or a["ws"] != self._wmap[p])) or (
                    st._absmax is not None and 1 - st._absmax_index != self._amap[p]):
                raise RuntimeError("SteadyLoop: buffer parity lost between eager and queued steps")
            self._after_eager = False
        # what changes per step: Adam's step number, the shuffle position of the NEXT batch
        e, b = divmod(k + 1, pipe.per_epoch)
        seed, first = ld.seed + 7919 * (pipe.epoch0 + e), ld.span(b)[0]
        a = self._args[p]
        self._check_pins(p)
        a.step, a.seed, a.first = opt.step_count + 1, seed & 0xFFFFFFFFFFFFFFFF, first
        a.lr = opt.param_groups[0]["lr"]
        a.join_pending = 1 if self._join_pending else 0
        a.stream = torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())
        evs = None
        if sample and st.phase_events is not None:
            pool = getattr(self, "_event_pool", None)
            if pool:
                evs = pool.pop()
            else:
                evs = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
                for ev in evs:
                    ev.record()  # materialises the handle (re-recorded inside the call)
            for i, ev in enumerate(evs):
                a.ev_phase[i] = ev.cuda_event
        _lib.call("mri_fused_step", C.byref(a))
        if evs is not None:
            for i, name in enumerate(self.PHASES):
                if st.world == 1 or name != "adam":  # (several ranks: Adam is queued below, behind the reduction)
                    st.phase_events.setdefault(name, []).append((evs[i], evs[i + 1]))
                a.ev_phase[i] = None
            a.ev_phase[4] = None
        self._join_pending = True
        if st.world > 1:  # data parallel, plain form: ONE reduction of the flat gradient, then Adam
            st.last_group_bytes = [st.flat.grad.numel() * 4]
            timed = sample and st.phase_events is not None
            saved, st.phase_events = st.phase_events, (st.phase_events if timed else None)
            with st._phase("all_reduce"), st._phase("reduce_wait_0"):
                parallel.all_reduce_sum(st.flat.grad)
            with st._phase("adam"):
                opt.step()
            st.phase_events = saved
            opt.step_count -= 1  # (counted below, as for one rank)
        # what an eager step would have left behind
        opt.step_count += 1
        pipe._made[(k + 1) % 2] = k + 1
        pipe.advance()
        st._ws_index = self._wmap[p]
        st._ahead = dict(ptr=pipe.slots[1 - p][1].data_ptr(), n=ld.batch_size, ws=self._wmap[1 - p],
                         event=None) if st.count_ahead else None
        st._counted = False
        if st._absmax is not None:
            st._absmax_index = self._amap[p]
            st._absmax_clean[self._amap[p]], st._absmax_clean[self._amap[1 - p]] = False, True
        return st.loss

    def finish(self):
        """Order the current stream behind everything queued (call before reading results or going eager)."""
        if self._join_pending:
            torch.cuda.current_stream().wait_event(self._ev_join)
            self._join_pending = False


class Trainer:
    """fit / predict with the subset of `pl.Trainer` the reference launcher uses."""

    def __init__(self, max_epochs: int = 1, max_steps: int = -1, accelerator: str = "gpu",
                 precision: int = 32, log_every: int = 0, distributed: bool = True,
                 accumulate_grad_batches=None, dp_mode: str = "all_reduce", grad_buckets: int = 1,
                 batch_group: int = 1, native_steps: bool = True):
        """`accumulate_grad_batches`: an int k (gradients of k consecutive batches are summed,
        each scaled by 1/k, before one Adam step -- what `pl.Trainer(accumulate_grad_batches=k)`
        does, reference launcher.py:159-161) or a mapping {epoch: k} (k from that epoch on, the
        scheduler form the reference's config holds, config/base.py:27).  Like Lightning, an
        epoch's last batch always steps."""
        if precision != 32:
            raise ValueError("the MI355X path is fp32 only (BASELINE parity is 1e-5 fp32)")
        if accelerator not in ("gpu", "auto", "cuda"):
            raise ValueError(f"accelerator {accelerator!r}: the hot path runs on the MI355X only")
        self.max_epochs, self.max_steps, self.log_every = max_epochs, max_steps, log_every
        self.accumulate = _accumulate_schedule(accumulate_grad_batches)
        if dp_mode not in ("all_reduce", "reduce_scatter"):
            raise ValueError(f"dp_mode {dp_mode!r} not in ('all_reduce', 'reduce_scatter')")
        self.dp_mode = dp_mode  # how a data-parallel FusedStep exchanges gradients, see there
        self.grad_buckets = int(grad_buckets)  # level groups of the table gradient's reduction, see there
        self.batch_group = int(batch_group)    # batches per launch of the on-device producer (BatchPipeline)
        self.native_steps = bool(native_steps)  # queue steady-state steps with one library call (SteadyLoop)
        self.rank, self.world = 0, 1
        if distributed:
            rank, world, _ = parallel.env_world()
            if world > 1 and parallel.world_size() != world:
                # FusedStep pre-divides gradients by the world size: without a process group
                # that would silently train on 1/world of the gradient
                raise RuntimeError(f"WORLD_SIZE={world} but no process group of that size is "
                                   "initialised: call parallel.init() first (or pass "
                                   "distributed=False)")
            self.rank, self.world = rank, world
        self.global_step = 0
        self.history: List[float] = []
        self.throughput: List[float] = []
        self.fused: Optional[FusedStep] = None

    def _prepare(self, model):
        if next(model.parameters()).device.type != "cuda":
            model.cuda()
        opt = getattr(model, "optimizer", None)
        if not isinstance(opt, optim.Adam):
            opt = model.configure_optimizers()
            model.optimizer = opt
        try:
            self.fused = FusedStep(model, opt, self.world)
        except ValueError:
            self.fused = None
            opt.flatten()
        if self.fused is not None:  # (a refused dp_mode is the caller's error, not "model does not fuse")
            self.fused.dp_mode = self.dp_mode
            self.fused.grad_buckets = max(1, self.grad_buckets)

        return opt

    def _check_equal_steps(self, loader, device):
        """Every rank must run the same number of steps (each one holds a collective)."""
        if self.world <= 1:
            return
        n = float(len(loader))
        hi = parallel.all_reduce_max(n, device)
        lo = -parallel.all_reduce_max(-n, device)
        if hi != lo:
            raise RuntimeError(f"rank {self.rank}: {int(n)} batches per epoch, other ranks "
                               f"{int(lo)}..{int(hi)}: ranks would leave the gradient all-reduce "
                               "at different steps; build the loaders with "
                               "datamodules.sharded_loader() (equal step counts and batch sizes)")

    def fit(self, model, train_dataloaders):
        opt = self._prepare(model)
        model.train()
        done = False
        self._check_equal_steps(train_dataloaders, next(model.parameters()).device)
        # fused path over an on-device loader: ONE double-buffered pipeline for all epochs, the
        # next batch (also the first one of the next epoch) is produced while a step runs
        pipe = None
        if self.fused is not None and isinstance(train_dataloaders, DeviceLoader) \
                and len(train_dataloaders) > 0:
            train_dataloaders.set_epoch(0)
            pipe = BatchPipeline(train_dataloaders, group=self.batch_group)
            self.fused.forget_ahead()  # a new pipeline: nothing counted earlier is about its batches
        # one library call per step (SteadyLoop, native form) once the first eager steps have set up the
        # steady state -- same launches on the same data, bit-identical parameters
        loop, eager_done = None, 0
        if pipe is not None and self.native_steps and SteadyLoop.unsupported(self.fused, pipe) is None \
                and all(k == 1 for k in self.accumulate.values()):
            loop = SteadyLoop(self.fused, pipe, mode="native")
        for epoch in range(self.max_epochs):
            if pipe is None and hasattr(train_dataloaders, "set_epoch"):
                train_dataloaders.set_epoch(epoch)
            t0, seen = time.perf_counter(), 0
            k_acc = _accumulate_at(self.accumulate, epoch)
            micro = 0  # position inside the current accumulation group
            n_batches = len(train_dataloaders)
            if pipe is not None:
                batches = ((b,) + pipe.current() for b in range(len(train_dataloaders)))
            else:
                batches = ((b, x, y) for b, (x, y) in enumerate(train_dataloaders))
            for batch_idx, x, y in batches:
                # accumulation group: the first micro-batch starts the gradient, the last one
                # (or the epoch's last batch) reduces it and steps
                stepping = micro == k_acc - 1 or batch_idx == n_batches - 1
                group = dict(first=micro == 0, step=stepping, divisor=float(k_acc))
                if pipe is not None:
                    final = (epoch == self.max_epochs - 1 and batch_idx == n_batches - 1) \
                        or (stepping and 0 < self.max_steps <= self.global_step + 1)
                    if loop is not None and eager_done >= 4 and not final:
                        if loop._wmap is None:
                            loop.capture(warm_steps=0)
                        loss = loop.step_once()
                    else:
                        if loop is not None:
                            loop.finish()
                            loop._after_eager = True
                        loss = self.fused.train_step(x, y, None if final else pipe.produce_next,
                                                     late_work=None if final else pipe.produce_late, **group)
                        eager_done += 1
                        if not final:
                            pipe.advance()
                elif self.fused is not None:
                    loss = self.fused.train_step(x, y, **group)
                else:
                    if micro == 0:
                        opt.zero_grad()
                    loss = model.training_step((x, y), batch_idx)
                    (loss / (self.world * k_acc)).backward()
                    if stepping:
                        if self.world > 1:
                            parallel.all_reduce_sum(opt.flatten().grad)
                        opt.step()
                micro = 0 if stepping else micro + 1
                if stepping:  # as Lightning counts: optimizer steps
                    self.global_step += 1
                seen += x.shape[0]
                if self.log_every and stepping and self.global_step % self.log_every == 0:
                    self.history.append(float(loss))
                    if self.rank == 0:
                        print(f"epoch {epoch} step {self.global_step} loss {self.history[-1]:.6e}",
                              flush=True)
                if stepping and 0 < self.max_steps <= self.global_step:
                    done = True
                    break
            if loop is not None:
                loop.finish()
            torch.cuda.synchronize()
            if self.fused is not None and self.fused.overlap_forward:
                self.fused.check_status()  # a decoder workgroup that gave up waiting invalidates the epoch
            self.throughput.append(seen * self.world / max(time.perf_counter() - t0, 1e-9))
            if done:
                break
        return model

    @torch.no_grad()
    def predict(self, model, dataloaders) -> List[torch.Tensor]:
        """List of per-batch predictions, as `pl.Trainer.predict` returns."""
        if next(model.parameters()).device.type != "cuda":
            model.cuda()
        model.eval()
        fused = self.fused
        if fused is None:
            try:
                fused = FusedStep(model, model.optimizer if hasattr(model, "optimizer")
                                  else model.configure_optimizers(), 1)
            except ValueError:
                fused = None
        out = []
        for batch_idx, (x, y) in enumerate(dataloaders):
            if fused is not None:
                pred, _ = fused.forward(x, train=False)
                out.append(pred.clone())
            else:
                out.append(model.predict_step((x, y), batch_idx))
        return out


def level_group_ranges(sizes, groups: int):
    """Level ranges [(lo, hi)] of a bucketed table gradient, in execution order.

    Every level costs the same to compute (same number of corners) but its all-reduce costs its
    table size, and only the LAST group's reduction cannot hide behind later compute.  So the
    full-size (hashed) levels are cut into groups - 1 contiguous groups that run first, finest
    first, and the coarse levels -- a few per cent of the bytes -- go last."""
    n_levels = len(sizes)
    groups = max(1, min(groups, n_levels))
    first_full = n_levels
    while first_full > 0 and sizes[first_full - 1] == max(sizes):
        first_full -= 1
    if groups >= 2 and 0 < first_full and n_levels - first_full >= groups - 1:
        k = groups - 1
        cuts = [first_full + round(g * (n_levels - first_full) / k) for g in range(k + 1)]
        return [(lo, hi) for lo, hi in zip(cuts, cuts[1:])][::-1] + [(0, first_full)]
    cuts = [round(g * n_levels / groups) for g in range(groups + 1)]
    return [(lo, hi) for lo, hi in zip(cuts, cuts[1:])][::-1]


def gradient_group_slices(sizes, n_features: int, table_offset: int, flat_numel: int,
                          groups: int):
    """[(level mask or None, lo, hi)]: the slices of the flat gradient buffer a data-parallel
    step reduces, in the order their reductions start -- what is not the hash table first (the
    decoder's gradients are complete before the table gradient starts), then the level groups.
    The slices cover [0, flat_numel) exactly once (checked by FusedStep on every step)."""
    rows = [0]
    for t in sizes:
        rows.append(rows[-1] + int(t))
    t0, t1 = table_offset, table_offset + rows[-1] * n_features
    out = []
    if t0 > 0:
        out.append((None, 0, t0))
    if t1 < flat_numel:
        out.append((None, t1, flat_numel))
    for lo, hi in level_group_ranges(list(sizes), groups):
        mask = sum(1 << l for l in range(lo, hi))
        out.append((mask, t0 + rows[lo] * n_features, t0 + rows[hi] * n_features))
    return out


def covers_exactly_once(slices, numel: int) -> bool:
    spans = sorted((lo, hi) for _, lo, hi in slices)
    return bool(spans) and spans[0][0] == 0 and spans[-1][1] == numel and \
        all(a[1] == b[0] for a, b in zip(spans, spans[1:])) and all(lo < hi for lo, hi in spans)


def _accumulate_schedule(spec):
    """{first epoch: batches per step}, sorted; None / 1 -> {0: 1}."""
    if spec is None:
        return {0: 1}
    if isinstance(spec, int):
        sched = {0: spec}
    else:
        try:
            sched = {int(e): int(k) for e, k in dict(spec).items()}
        except (TypeError, ValueError):
            raise TypeError(f"accumulate_grad_batches must be an int or a mapping epoch -> int, "
                            f"got {spec!r}") from None
    if any(k < 1 for k in sched.values()) or any(e < 0 for e in sched):
        raise ValueError(f"accumulate_grad_batches {spec!r}: counts must be >= 1, epochs >= 0")
    sched.setdefault(0, 1)
    return dict(sorted(sched.items()))


def _accumulate_at(sched, epoch: int) -> int:
    k = 1
    for e, v in sched.items():
        if e <= epoch:
            k = v
    return k


def psnr(pred: torch.Tensor, target: torch.Tensor) -> float:
    """10 log10(1 / MSE) for intensities in [0, 1] (what
    skimage.metrics.peak_signal_noise_ratio computes for non-negative float images,
    reference legacy_code/hash_experimentation.py:445-450)."""
    mse = torch.mean((pred.double() - target.double()) ** 2)
    return float(10.0 * torch.log10(1.0 / mse))
