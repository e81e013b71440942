"""Training engine: the role Keras' ``compile / fit_generator / evaluate_generator /
predict_generator`` + ``multi_gpu_model`` play in the reference (learn_image_embeddings.py:120-148,
224-255,270-275), re-designed for MI355X:

* one process per GPU; gradients are all-reduced with RCCL (``torch.distributed`` backend "nccl")
  in contiguous buckets that are launched from autograd hooks while backward is still running;
* all parameters / gradients / momentum buffers live in three flat fp32 HBM buffers, so the
  Keras-style update (L2 regulariser folded into the gradient BEFORE clipping, global-norm
  clipping, ``lr / (1 + decay * t)``, momentum / Nesterov) is a handful of whole-buffer launches
  regardless of the number of layers;
* the loss head is the fused HIP kernel (``utils.CosineEmbeddingLoss``); the backbone (MIOpen / hipBLASLt) runs in the
  mode ``backbone_mode`` picks per architecture -- fp32 NCHW for the CIFAR-sized ResNets (their 16-64 channel
  convolutions are launch-bound: bf16 autocast only adds ~880 cast/accumulate launches per step), bf16 autocast
  channels_last for the ImageNet-sized ones; master weights and the loss are always fp32;
* launch-bound steps are replayed as HIP graphs (``Trainer.enable_graphs``), validated against the eager gradient
  at capture time;
* metrics are accumulated on the device and read back once per epoch (no per-step host sync).

Semantics mirrored from Keras 2.2 [third party, not in the reference tree]: SGD velocity
``v = m v - lr g; w += v`` (Nesterov: ``w += m v - lr g``), ``clipnorm`` = global-norm clip,
per-replica BatchNorm statistics (``multi_gpu_model`` towers == no SyncBN).
"""
import os
import time

import numpy as np
import torch
import torch.distributed as dist


def dist_info():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def backbone_mode(architecture):
    """``(autocast_dtype, memory_format)`` for ``Trainer``: how the PyTorch-ROCm backbone of ``architecture`` runs.
    Measured on MI355X, resnet-110-fc, batch 128 (tools/train_variants.py, tools/graph_variants.py): bf16 autocast
    channels_last 31.5 ms/step, fp32 channels_last 23.7, fp32 NCHW 21.6 (eager; all host-launch-bound), fp32 NCHW
    HIP-graph replay 15.3.  ``SE_TRAIN_DTYPE`` = fp32|bf16 and ``SE_TRAIN_LAYOUT`` = nchw|nhwc override."""
    small = str(architecture).startswith(('resnet-32', 'resnet-110', 'simple', 'wrn-'))      # CIFAR-sized inputs
    dtype = os.environ.get('SE_TRAIN_DTYPE', 'fp32' if small else 'bf16').lower()
    layout = os.environ.get('SE_TRAIN_LAYOUT', 'nchw' if small else 'nhwc').lower()
    if dtype not in ('fp32', 'bf16') or layout not in ('nchw', 'nhwc'):
        raise ValueError('SE_TRAIN_DTYPE must be fp32|bf16 and SE_TRAIN_LAYOUT nchw|nhwc')
    return (None if dtype == 'fp32' else torch.bfloat16,
            torch.contiguous_format if layout == 'nchw' else torch.channels_last)


FLAT_ALIGN_ELEMS = 64      # every parameter slice of the flat buffers starts on a 256-byte boundary


class FlatState(object):
    """Re-homes every trainable parameter of ``model`` into one flat fp32 buffer (keeping each
    parameter's shape and strides, e.g. channels_last conv kernels), with matching flat gradient,
    velocity and L2-coefficient buffers.

    Every slice starts on a 256-byte boundary (``align`` elements; the padding words stay zero in all four buffers, so the
    whole-buffer update, clip norm and all-reduce see zeros there).  A freshly allocated parameter tensor is 256-byte aligned
    in PyTorch and vendor kernels may assume so; packing 16-64-element bias vectors back to back put them at arbitrary 4-byte
    offsets (round 4 excluded this as the cause of the bf16 replay fault, round 5 found the cause elsewhere: DESIGN.md section 7.1)."""

    def __init__(self, model, l2_of=None, align=None):
        params = [p for p in model.parameters() if p.requires_grad]
        self.params = params
        align = int(os.environ.get('SE_FLAT_ALIGN', FLAT_ALIGN_ELEMS)) if align is None else int(align)
        self.align = max(align, 1)
        starts, off = [], 0
        for p in params:
            off = (off + self.align - 1) // self.align * self.align
            starts.append(off)
            off += p.numel()
        total = (off + self.align - 1) // self.align * self.align
        dev = params[0].device
        self.flat_p = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_v = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_l2 = torch.zeros(total, dtype=torch.float32, device=dev)
        self.offsets = []
        l2_of = l2_of or {}
        for p, off in zip(params, starts):
            n = p.numel()
            dense = p.is_contiguous() or p.is_contiguous(memory_format=torch.channels_last)
            src = p.data if dense else p.data.contiguous()
            view = self.flat_p[off:off + n].as_strided(src.shape, src.stride())
            view.copy_(src)
            p.data = view
            p.grad = self.flat_g[off:off + n].as_strided(src.shape, src.stride())
            lam = l2_of.get(id(p), 0.0)
            if lam:
                self.flat_l2[off:off + n] = 2.0 * lam        # d/dw (lam * w^2)
            self.offsets.append((off, n))
        self.total = total
        self.frozen_l2 = [(p, float(l2_of[id(p)])) for p in model.parameters() if not p.requires_grad and l2_of.get(id(p), 0.0)]
        self.has_l2 = bool((self.flat_l2 != 0).any().item())
        self.all_contiguous = all(p.is_contiguous() for p in params)
        self.packed = all(o == (self.offsets[i - 1][0] + self.offsets[i - 1][1] if i else 0) for i, (o, _) in enumerate(self.offsets)) \
            and total == sum(n for _, n in self.offsets)
        self._grad_views = [self.flat_g[o:o + n].view(p.shape) for p, (o, n) in zip(params, self.offsets)] if self.all_contiguous else None

    def bind_grads(self):
        """(Re-)point every ``p.grad`` at its slice of the flat gradient buffer (accumulating mode)."""
        for p, (off, n) in zip(self.params, self.offsets):
            p.grad = self.flat_g[off:off + n].as_strided(p.shape, p.stride())

    def gather_grads(self):
        """Stolen-gradient mode: autograd left every gradient in a tensor of its own (``p.grad`` was None, so nothing
        was accumulated); pack them into the flat buffer with one batched copy (a handful of multi-tensor launches for 442
        tensors; the padding words between the slices are never written and stay zero)."""
        grads = [(p.grad if p.grad is not None else torch.zeros_like(p)) for p in self.params]
        if self.packed:
            torch.cat([g.reshape(-1) for g in grads], out=self.flat_g)
        else:
            torch._foreach_copy_(self._grad_views, grads)


class BucketedAllReduce(object):
    """Gradient all-reduce over RCCL in contiguous buckets of the flat gradient buffer, launched
    as soon as every gradient of a bucket has been accumulated (post-accumulate-grad hooks), i.e.
    overlapped with the rest of backward.  Buckets are cut in reverse parameter order because
    autograd produces the last layers' gradients first."""

    def __init__(self, flat, bucket_bytes=25 << 20):
        self.flat = flat
        self.rank, self.world = dist_info()
        self.enabled = self.world > 1
        self.handles = []
        if not self.enabled:
            return
        cap = max(1, bucket_bytes // 4)
        self.bucket_of = {}
        self.buckets = []              # (start, end, n_params)
        end = flat.total
        count, start = 0, end
        for idx in range(len(flat.params) - 1, -1, -1):
            off, n = flat.offsets[idx]
            start = off
            count += 1
            self.bucket_of[idx] = len(self.buckets)
            if end - start >= cap or idx == 0:
                self.buckets.append([start, end, count])
                end, count = start, 0
        self.pending = [b[2] for b in self.buckets]
        self.hook_handles = [p.register_post_accumulate_grad_hook(self._make_hook(idx)) for idx, p in enumerate(flat.params)]

    def close(self):
        """Detach from the parameters (a second Trainer over the same model -- e.g. after the --finetune warm-up -- must not
        inherit hooks that all-reduce this one's dead gradient buffer)."""
        for h in getattr(self, 'hook_handles', ()):
            h.remove()
        self.hook_handles = []
        for h in self.handles:
            h.wait()
        self.handles = []
        self.enabled = False

    def _make_hook(self, idx):
        def hook(param):
            if not self.enabled:      # graph mode: one eager all-reduce between the captured graphs instead
                return
            b = self.bucket_of[idx]
            self.pending[b] -= 1
            if self.pending[b] == 0:
                s, e, _ = self.buckets[b]
                self.handles.append(dist.all_reduce(self.flat.flat_g[s:e], op=dist.ReduceOp.SUM, async_op=True))
        return hook

    def finish(self):
        """Wait for the in-flight buckets; returns the factor that turns the summed gradient into
        the mean over ranks (applied inside the fused update)."""
        if not self.enabled:
            return 1.0
        for b, left in enumerate(self.pending):
            if left != 0:   # parameters that received no gradient this step (frozen / unused)
                s, e, _ = self.buckets[b]
                self.handles.append(dist.all_reduce(self.flat.flat_g[s:e], op=dist.ReduceOp.SUM, async_op=True))
        for h in self.handles:
            h.wait()
        self.handles = []
        self.pending = [b[2] for b in self.buckets]
        return 1.0 / self.world


class Trainer(object):
    """``compile`` + ``fit`` in one object.

    losses:  dict output_name -> (loss_fn(y_true, y_pred) -> [B], weight); outputs of the model
             are matched by position (tuple outputs) in the dict's order.
    metrics: dict output_name -> list of metric fns with ``.name``.
    A batch from the sequences is ``(X, y)`` or ``(X, [y_1, y_2, ...])``."""

    def __init__(self, model, losses, metrics=None, lr=0.1, momentum=0.9, nesterov=False, clipnorm=None, decay=0.0,
                 l2_of=None, autocast_dtype=torch.bfloat16, bucket_bytes=25 << 20, trainable=None, memory_format=None):
        self.model = model
        self.memory_format = memory_format       # None: leave model and batches as they come (channels_last by construction)
        if memory_format is not None:
            model.to(memory_format=memory_format)
        self.losses = losses
        self.metrics = metrics or {}
        self.lr, self.momentum, self.nesterov, self.clipnorm, self.decay = lr, momentum, nesterov, clipnorm, decay
        self.autocast_dtype = autocast_dtype
        self.iterations = 0
        self.rank, self.world = dist_info()
        self.is_main_process = self.rank == 0
        if trainable is not None:
            for name, p in model.named_parameters():
                p.requires_grad_(trainable(name))
        self.flat = FlatState(model, l2_of)
        self.reducer = BucketedAllReduce(self.flat, bucket_bytes)
        self.stop_training = False
        self._graph = None
        self._graph_tried = False
        self._grads_bound = True           # p.grad are views of flat_g (FlatState); False while gradients are stolen

    # ---------------------------------------------------------------- one step

    def _forward(self, X):
        if self.memory_format is not None and X.dim() == 4:
            X = X.contiguous(memory_format=self.memory_format)     # no-op when the loader already produces this layout
        if self.autocast_dtype is not None and X.is_cuda:
            # cache_enabled=False: the weight-cast cache keeps tensors that were created inside a HIP-graph capture alive across
            # replays (every weight is used once per forward here, so the cache saves nothing anyway)
            with torch.autocast('cuda', dtype=self.autocast_dtype, cache_enabled=False):
                out = self.model(X)
        else:
            out = self.model(X)
        return out if isinstance(out, (tuple, list)) else (out,)

    def _loss_and_metrics(self, outs, y, logs):
        ys = y if isinstance(y, (tuple, list)) else (y,)
        total = None
        for (name, (fn, weight)), out, yt in zip(self.losses.items(), outs, ys):
            li = fn(yt, out)
            # a fused loss head exposes the normalised embedding it computed; metrics are defined on it
            m_in = getattr(fn, 'last_normalized', None)
            m_in = out.detach() if m_in is None else m_in
            mean = li.mean()
            total = mean * weight if total is None else total + mean * weight
            key = 'loss' if len(self.losses) == 1 else name + '_loss'
            nrows = int(li.shape[0])          # logs accumulate SUMS over samples (Keras weights every batch by its size); '_n' counts them
            logs[key] = logs.get(key, 0) + mean.detach() * nrows
            for m in self.metrics.get(name, ()):
                mname = m.name if len(self.losses) == 1 else '{}_{}'.format(name, m.name)
                with torch.no_grad():
                    logs[mname] = logs.get(mname, 0) + m(yt, m_in).float().sum()
        if len(self.losses) > 1:
            logs['loss'] = logs.get('loss', 0) + total.detach() * nrows
        logs['_n'] = logs.get('_n', 0) + nrows
        return total

    def train_step(self, X, y, logs):
        if self._graph is not None:
            return self._graph_step(X, y, logs)
        loss = self._eager_core(X, y, logs)
        scale = self.reducer.finish()
        self.apply_update(scale)
        return loss

    # ---------------------------------------------------------------- HIP-graph replay of the step

    def enable_graphs(self, X, y, warmup=3, validate=4, tol=2e-3, max_noise=0.05, allow_autocast=False):
        """Capture the training step into two HIP graphs (``torch.cuda.CUDAGraph``): A = zero grads + forward + fused
        loss/metric + backward, B = the whole-buffer SGD update; the RCCL all-reduce of the flat gradient buffer stays
        an eager call between them (one message per step), so 1-GPU and N-GPU runs replay identical graphs.
        A ResNet-110 step is ~1500 small launches and host-launch-bound in eager mode (21.6 -> 15.3 ms on MI355X, fp32).
        Batches of another shape than ``(X, y)`` (a short last batch) run eagerly.

        The model's state is untouched: parameters, velocity and BatchNorm buffers are snapshotted before the warm-up
        steps the capture needs (MIOpen / hipBLASLt algorithm searches) and restored afterwards.  Graph A is then
        **validated**: ``validate`` replays must reproduce the eager gradient of the same batch -- relative L2 error and
        norm ratio within ``max(tol, 2 x the eager step's own run-to-run spread)``, capped at ``2 max_noise``; the loss within
        1e-3; everything finite.  A step whose eager gradients already differ by more than ``max_noise`` between two runs
        cannot be validated and stays eager.  fp32 steps only (bf16-autocast replays are not offered).
        On failure the trainer stays eager, says so, and returns False."""
        if not X.is_cuda:
            return False
        if self.autocast_dtype is not None and not allow_autocast:
            # bf16 replays of these backbones return wrong conv-bias gradients -- not because of this trainer (a bare torch.cuda.graph of
            # the network fails the same way): one aten::convolution_backward on bf16 channels_last tensors (MIOpen's split-K weight
            # gradient) depends on memory outside the graph's pool when replayed: its result changes with what the allocator recycled in
            # between (tools/graph_wrw_bf16_repro.py).  The mode is not offered; it would save nothing either
            # (the bf16 ResNet-50 step is GPU-bound in eager mode, bf16 is slower than fp32 replay for the CIFAR nets).
            print('[engine] HIP-graph replay is offered for fp32 steps only; staying eager', flush=True)
            return False
        ys = y if isinstance(y, (tuple, list)) else (y,)
        hooks_were = self.reducer.enabled
        snap_p, snap_v, snap_it = self.flat.flat_p.clone(), self.flat.flat_v.clone(), self.iterations
        snap_buf = [b.clone() for b in self.model.buffers()]

        def restore():
            self.flat.flat_p.copy_(snap_p)
            self.flat.flat_v.copy_(snap_v)
            for b, sb in zip(self.model.buffers(), snap_buf):
                b.copy_(sb)
            self.iterations = snap_it

        try:
            self._sX = X.clone()
            self._sy = [t.clone() for t in ys]
            sy = self._sy if len(self._sy) > 1 else self._sy[0]
            self._lr_t = torch.full((), float(self.lr), dtype=torch.float32, device=X.device)
            self.reducer.enabled = False          # no collectives inside the capture: all-reduce runs between the graphs
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(warmup):           # MIOpen / hipBLASLt algorithm searches must happen before the capture
                    self._eager_core(self._sX, sy, {})
                    self.apply_update(1.0, lr_tensor=self._lr_t, count=False)
            torch.cuda.current_stream().wait_stream(side)
            restore()
            self._g_logs = {}
            ga = torch.cuda.CUDAGraph()
            # multi-process: the RCCL watchdog thread polls events while we capture; only this thread's calls may invalidate the capture
            mode = dict(capture_error_mode='thread_local') if self.world > 1 else {}
            with torch.cuda.graph(ga, **mode):
                self._g_loss = self._eager_core(self._sX, sy, self._g_logs)
            gb = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gb, **mode):
                self.apply_update(1.0 / self.world if hooks_were else 1.0, lr_tensor=self._lr_t, count=False)
            if validate:
                # The yardstick is the eager step's OWN run-to-run spread on this batch: MIOpen's kernels are not bit-reproducible, and a
                # deep randomly initialised network amplifies their rounding noise (measured, tools/grad_noise_probe.py: two eager
                # backward passes of the bf16 ResNet-50 differ by 0.7 relative L2 in every layer, 7e-3 in fp32; the CIFAR ResNets 1e-5 /
                # 1e-3).  A replay must be as close to an eager gradient as a second eager gradient is, have the same norm and loss.
                self._eager_core(self._sX, sy, {})
                ref = self.flat.flat_g.clone()
                eager_logs = {}
                self._eager_core(self._sX, sy, eager_logs)
                ref_norm = float(torch.linalg.vector_norm(ref))
                noise = float(torch.linalg.vector_norm(self.flat.flat_g - ref)) / max(ref_norm, 1e-30)
                eager_loss = float(eager_logs['loss']) / max(float(eager_logs.get('_n', 1)), 1.0) if 'loss' in eager_logs else float(self._g_loss)
                self.graph_validation = {'eager_noise': noise, 'replay_error': []}
                if not noise < max_noise:
                    # two eager gradients of the same batch differ by more than `max_noise` (relative L2): no tolerance derived from
                    # that spread could tell a right replay from a wrong one (two unrelated gradients of equal norm differ by 1.41)
                    raise RuntimeError('eager gradients of one batch differ by %g between two runs: a replay cannot be validated' % noise)
                bound = min(max(tol, 2.0 * noise), 2.0 * max_noise)
                for r in range(validate):
                    ga.replay()
                    g = self.flat.flat_g
                    err = float(torch.linalg.vector_norm(g - ref)) / max(ref_norm, 1e-30)
                    ratio = float(torch.linalg.vector_norm(g)) / max(ref_norm, 1e-30)
                    self.graph_validation['replay_error'].append(err)
                    ok = (err < bound) and (abs(ratio - 1.0) < bound) and bool(torch.isfinite(g).all()) \
                        and bool(torch.isfinite(self._g_loss)) and abs(float(self._g_loss) - eager_loss) < 1e-3 * max(1.0, abs(eager_loss))
                    if not ok:      # (NaN compares False)
                        raise RuntimeError('replay %d of the captured step does not reproduce the eager gradient (relative L2 error %g, '
                                           'bound %g; %g between two eager steps; norm ratio %g, loss %g vs %g)'
                                           % (r, err, bound, noise, ratio, float(self._g_loss), eager_loss))
            restore()
            graphs, why = (ga, gb), None
        except Exception as e:
            graphs, why = None, '%s: %s' % (type(e).__name__, e)
            torch.cuda.synchronize()
            restore()
        if self.world > 1:                        # every rank must take the same path (graph mode changes the collective pattern)
            agree = torch.tensor([1 if graphs is not None else 0], dtype=torch.int32, device=X.device)
            dist.all_reduce(agree, op=dist.ReduceOp.MIN)
            if int(agree.item()) == 0 and graphs is not None:
                graphs, why = None, 'another rank failed the capture or its validation'
        if graphs is None:                        # stay eager, loudly
            print('[engine] HIP-graph replay disabled, staying eager: %s' % why, flush=True)
            self._graph = None
            self.reducer.enabled = hooks_were
            return False
        self._graph = graphs
        self._graph_allreduce = hooks_were
        self._graph_steps = 0
        return True

    def close(self):
        """Release what this trainer attached to the model (gradient hooks); the model keeps its current weights."""
        self.reducer.close()
        self._graph = None

    def _eager_core(self, X, y, logs):
        """Zero / forward / loss + metrics / backward; leaves the (rank-local) gradient in ``flat.flat_g``.
        Without bucket hooks to feed (single process, or graph mode where the all-reduce is one call after backward) the
        gradients are *stolen*: ``p.grad = None`` makes autograd hand over each gradient tensor as it is instead of
        launching one ``flat_g[slice] += grad`` per parameter (442 launches, 1.5 ms of GPU time on resnet-110-fc), and one
        batched ``cat`` packs them into the flat buffer."""
        flat = self.flat
        steal = flat.all_contiguous and not self.reducer.enabled
        if steal:
            for p in flat.params:
                p.grad = None
            self._grads_bound = False
        else:
            if not self._grads_bound:
                flat.bind_grads()
                self._grads_bound = True
            flat.flat_g.zero_()
        outs = self._forward(X)
        loss = self._loss_and_metrics(outs, y, logs)
        loss.backward()
        if steal:
            flat.gather_grads()
        return loss.detach()

    def _allreduce_flat(self, on):
        if on and self.world > 1:
            dist.all_reduce(self.flat.flat_g, op=dist.ReduceOp.SUM)

    def _graph_step(self, X, y, logs):
        ys = y if isinstance(y, (tuple, list)) else (y,)
        if X.shape != self._sX.shape or any(a.shape != b.shape for a, b in zip(self._sy, ys)):
            # e.g. the short last batch of an epoch: same arithmetic, launched eagerly (hooks are off in graph mode)
            loss = self._eager_core(X, y, logs)
            self._allreduce_flat(self._graph_allreduce)
            self.apply_update(1.0 / self.world if self._graph_allreduce else 1.0)
            return loss
        self._sX.copy_(X, non_blocking=True)
        for dst, src in zip(self._sy, ys):
            dst.copy_(src, non_blocking=True)
        lr = self.lr / (1.0 + self.decay * self.iterations) if self.decay > 0 else self.lr
        self._lr_t.fill_(float(lr))
        ga, gb = self._graph
        ga.replay()
        self._allreduce_flat(self._graph_allreduce)
        gb.replay()
        self.iterations += 1
        self._graph_steps += 1
        for k, v in self._g_logs.items():      # static device scalars written by graph A (sums over the batch) and its row count
            logs[k] = logs.get(k, 0) + (v.clone() if torch.is_tensor(v) else v)
        if self._graph_steps % 256 == 0 and not (bool(torch.isfinite(self._g_loss)) and bool(torch.isfinite(self.flat.flat_p).all())):
            # one host sync per 256 steps.  Weights and loss are checked (the gradient buffer was already consumed by graph B).
            raise FloatingPointError('non-finite loss / weights in HIP-graph replay %d (diverged training, or a replay fault: '
                                     'rerun with SE_TRAIN_GRAPHS=0 to tell)' % self._graph_steps)
        return self._g_loss

    def apply_update(self, grad_scale=1.0, lr_tensor=None, count=True):
        """Keras-SGD update on the flat buffers (regulariser -> clip -> decayed lr -> momentum).  With ``lr_tensor``
        (a device scalar) the learning rate is read on the device, so the launches can be captured in a HIP graph."""
        flat = self.flat
        g = flat.flat_g
        if grad_scale != 1.0:
            g.mul_(grad_scale)
        if flat.has_l2:
            g.addcmul_(flat.flat_l2, flat.flat_p)
        if self.clipnorm:
            norm = torch.linalg.vector_norm(g)
            g.mul_(torch.clamp(self.clipnorm / (norm + 1e-12), max=1.0))   # stays on the device
        v = flat.flat_v
        if lr_tensor is not None:
            step = g * lr_tensor                      # lr lives on the device (graph replay)
            v.mul_(self.momentum).sub_(step)
            if self.nesterov:
                flat.flat_p.add_(v, alpha=self.momentum).sub_(step)
            else:
                flat.flat_p.add_(v)
        else:
            lr = self.lr / (1.0 + self.decay * self.iterations) if self.decay > 0 else self.lr
            v.mul_(self.momentum).add_(g, alpha=-lr)
            if self.nesterov:
                flat.flat_p.add_(v, alpha=self.momentum).add_(g, alpha=-lr)
            else:
                flat.flat_p.add_(v)
        if count:
            self.iterations += 1

    # ---------------------------------------------------------------- loops

    def regularization_loss(self):
        """Sum of the L2 kernel penalties (``lam * |w|^2``) at the current weights -- Keras adds them to every reported ``loss`` /
        ``val_loss`` (what ReduceLROnPlateau and --snapshot_best observe)."""
        flat = self.flat
        reg = 0.0
        if flat.has_l2:
            reg += float(0.5 * torch.dot(flat.flat_l2, flat.flat_p * flat.flat_p))     # flat_l2 holds 2 lam
        if flat.frozen_l2:                 # layers frozen by --finetune_init keep their regularisers in Keras' loss
            # their weights do not move while they stay frozen: one device reduction + ONE host sync per (frozen set, weight version)
            # instead of a launch and a sync per tensor on every call (~160 for a --finetune_init ResNet-50)
            key = tuple((id(p), p._version, bool(p.requires_grad)) for p, _ in flat.frozen_l2)
            if getattr(self, '_frozen_reg_key', None) != key:
                total = torch.zeros((), dtype=torch.float32, device=flat.flat_p.device)
                for p, lam in flat.frozen_l2:
                    total = total + lam * (p.detach().float() ** 2).sum()
                self._frozen_reg, self._frozen_reg_key = float(total), key
            reg += self._frozen_reg
        return reg

    def _reduce_logs(self, logs, n=None):
        """Per-sample means of the accumulated sums (``logs['_n']`` samples on this rank), summed over the ranks: every sample
        counts once whatever the batch / shard sizes.  ``loss`` gets the L2 penalties of the current weights added like Keras
        reports it (for the training loss of an epoch this is the penalty at the epoch's end, not its running average)."""
        out = {}
        keys = sorted(k for k in logs if k != '_n')
        if not keys:
            return out
        dev = self.flat.flat_p.device
        vec = torch.stack([torch.as_tensor(logs[k], dtype=torch.float32, device=dev) for k in keys]
                          + [torch.as_tensor(float(logs.get('_n', 0)), dtype=torch.float32, device=dev)])
        if self.world > 1:
            dist.all_reduce(vec, op=dist.ReduceOp.SUM)
        count = max(float(vec[-1].item()), 1.0)
        for k, v in zip(keys, (vec[:-1] / count).tolist()):
            out[k] = v
        if 'loss' in out:
            out['loss'] += self.regularization_loss()
        return out

    def evaluate(self, seq):
        self.model.eval()
        logs, n = {}, 0
        with torch.no_grad():
            for i in range(len(seq)):
                X, y = seq[i]
                self._loss_and_metrics(self._forward(X), y, logs)
                n += 1
        self.model.train()
        return self._reduce_logs(logs)

    def predict(self, seq, steps=None, to_host=True):
        """Model outputs for every batch of ``seq`` (rank-local rows), concatenated: NumPy arrays on the host (``to_host``, what
        the pickle dumps need) or float32 DEVICE tensors -- features go straight from the network into
        ``ClassHierarchy.hierarchical_precision_device`` / ``evaluate_retrieval.ranking_tiles`` without a host hop
        (learn_image_embeddings.py:270-275 -> evaluate_retrieval.py:43-54)."""
        self.model.eval()
        outs = None
        with torch.no_grad():
            for i in range(len(seq) if steps is None else steps):
                batch = seq[i]
                X = batch[0] if isinstance(batch, (tuple, list)) else batch
                o = [t.float().cpu() if to_host else t.float() for t in self._forward(X)]
                outs = [[t] for t in o] if outs is None else [a + [t] for a, t in zip(outs, o)]
        self.model.train()
        cat = [torch.cat(a).numpy() if to_host else torch.cat(a) for a in outs]
        return cat[0] if len(cat) == 1 else cat

    def fit(self, train_seq, validation_data=None, epochs=1, initial_epoch=0, callbacks=(), verbose=True, log_every=50):
        self.model.train()
        # steps are replayed as HIP graphs unless SE_TRAIN_GRAPHS=0; a capture that fails its validation against the eager
        # gradient (enable_graphs) leaves the trainer eager and says so
        if (not self._graph_tried and len(train_seq) > 0 and os.environ.get('SE_TRAIN_GRAPHS', '1') != '0'
                and self.autocast_dtype is None):
            self._graph_tried = True
            X0, y0 = train_seq[0]
            if X0.is_cuda:
                ok = self.enable_graphs(X0, y0)
                if verbose and self.is_main_process:
                    print('[engine] training step: %s' % ('HIP-graph replay' if ok else 'eager launches'), flush=True)
        for cb in callbacks:
            cb.on_train_begin(self)
        history = []
        for epoch in range(initial_epoch, epochs):
            for cb in callbacks:
                cb.on_epoch_begin(self, epoch)
            logs, t0, nb = {}, time.time(), len(train_seq)
            for b in range(nb):
                X, y = train_seq[b]
                self.train_step(X, y, logs)
                for cb in callbacks:
                    cb.on_batch_end(self, b, logs)
                if verbose and self.is_main_process and (b + 1) % log_every == 0:
                    print('\rEpoch {}/{} - batch {}/{} - {:.1f} img/s'.format(
                        epoch + 1, epochs, b + 1, nb, (b + 1) * train_seq.batch_size / (time.time() - t0)), end='', flush=True)
            train_seq.on_epoch_end()
            ep_logs = self._reduce_logs(logs)
            if validation_data is not None:
                ep_logs.update({'val_' + k: v for k, v in self.evaluate(validation_data).items()})
            for cb in callbacks:
                cb.on_epoch_end(self, epoch, ep_logs)
            history.append(ep_logs)
            if verbose and self.is_main_process:
                # keep the "name: value" format the reference's README greps (CosineLoss.md:95-104)
                print('\rEpoch {}/{} - {:.0f}s - '.format(epoch + 1, epochs, time.time() - t0) +
                      ' - '.join('{}: {:.4f}'.format(k, v) for k, v in ep_logs.items()), flush=True)
            if self.stop_training:
                break
        return history
