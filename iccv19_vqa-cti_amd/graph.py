"""hipGraph capture of a whole training step (reference src/FFOE/trainer.py:97-149,221-269).

The FFOE CTI training step is ~500 kernel launches issued by ~7 ms of host Python for ~7.7 ms of GPU work: the step is host-co-limited.
`GraphedTrainStep` captures the step ONCE (torch.cuda.CUDAGraph = hipGraph) and replays it per batch.  What makes the step capturable is that
nothing that changes from step to step is a kernel ARGUMENT: the learning rate, the completed-step count of Adamax' bias correction and the
dropout streams' step counter live in device memory (cti_adamax_step_g / cti_dropout_g / cti_counter_add), and the library allocates nothing
and never synchronises.

Two forms:
  * one rank, no collective: forward + loss + backward + FlatAdamaxDP.step() are ONE graph; the host issues one hipGraphLaunch per step;
  * with a process group (the data-parallel step of SURVEY 8e): capturing the RCCL all-reduce aborted the process on this stack (ROCm 7.0 /
    torch 2.10), so the step is TWO graphs with the collective issued eagerly between them --
        graph A: zero_grad, forward, loss, backward, cti_flat_gather          (everything up to the flat gradient buffer)
        eager  : dist.all_reduce(flat_g)                                      (the ONE collective, reference trainer.py:221-232)
        graph B: scale + norm, clip + Adamax, step / dropout counters
    = two hipGraphLaunch + one ncclAllReduce per step from the host instead of ~500 launches, so the ranks time the GPU and the xGMI, not Python.
Batches are fed by copying into static input tensors."""
import torch

from . import ops


class GraphedTrainStep:
    def __init__(self, model, optimizer, loss_fn, example_inputs, example_target, warmup=2, split_collective=None):
        """loss_fn(model_output, target) -> scalar tensor.  example_inputs: tuple of tensors (shapes / dtypes are frozen); warmup eager steps run on
        a side stream first (allocator, one-time attributes, weight-norm caches), as torch's capture rules ask.  The warm-up steps DO update the
        parameters.  split_collective: None = the two-graph form exactly when the optimizer issues a collective; True / False force it."""
        self.model, self.opt, self.loss_fn = model, optimizer, loss_fn
        self.split = optimizer.needs_collective() if split_collective is None else bool(split_collective)
        self.static_in = tuple(t.clone() for t in example_inputs)
        self.static_tgt = example_target.clone()
        self.graph = torch.cuda.CUDAGraph()
        self.graph_update = torch.cuda.CUDAGraph() if self.split else None
        # The warm-up steps run on ONE side stream, the capture on ANOTHER: with a process group the warm-up issues eager RCCL all-reduces whose
        # completion events live on the stream they ran on, and the process group's watchdog thread may still poll such an event after the
        # capture has begun -- HIP refuses a query of an event whose stream is capturing (hipErrorCapturedEvent) and the watchdog takes the
        # process down.  A stream that never captures keeps those events queryable.
        w = torch.cuda.Stream()
        w.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(w):
            for _ in range(warmup):
                self._step()
        torch.cuda.synchronize()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            ops.invalidate_caches()                      # the captured step must contain the per-step refresh of every derived cache
            if not self.split:
                with torch.cuda.graph(self.graph, stream=s):
                    self.static_loss = self._step()
            else:
                # (thread-local capture mode: the process group's watchdog thread keeps making HIP calls of its own while this thread captures)
                with torch.cuda.graph(self.graph, stream=s, capture_error_mode="thread_local"):
                    self.static_loss = self._forward_backward()
                with torch.cuda.graph(self.graph_update, stream=s, capture_error_mode="thread_local"):
                    self.opt.apply_update()
        torch.cuda.current_stream().wait_stream(s)

    def _forward_backward(self):
        self.opt.zero_grad()
        loss = self.loss_fn(self.model(*self.static_in), self.static_tgt)
        loss.backward()
        self.opt.gather_grads()
        return loss.detach()

    def _step(self):
        loss = self._forward_backward()
        self.opt.all_reduce_grads()
        self.opt.apply_update()
        return loss

    def replay(self):
        """The captured step on whatever the static tensors hold.  Replays rewrite the parameters on the device behind every host-side cache key
        (weight-norm scales, operand planes of TCNet / the GRU are keyed on a package-wide epoch + storage + autograd version, none of which a
        replay moves), so the epoch is bumped here: an eager forward between two replays must not reuse planes of the older parameters.
        Anyone replaying `.graph` directly must call ops.invalidate_caches() as well."""
        self.graph.replay()
        if self.split:
            self.opt.all_reduce_grads()                  # eager: the one collective, stream-ordered between the two graphs
            self.graph_update.replay()
        ops.invalidate_caches()

    def __call__(self, inputs, target):
        """One training step on a new batch (copied into the captured tensors); returns the loss tensor of the replay (device, no sync)."""
        for dst, src in zip(self.static_in, inputs):
            dst.copy_(src, non_blocking=True)
        self.static_tgt.copy_(target, non_blocking=True)
        self.replay()
        return self.static_loss


class GraphedForward:
    """hipGraph replay of an inference forward WITH the f16f6 range guard's safety net (VERDICT r5 #7a; the reference multiplies in full-range fp32,
    src/Tensor.py:12-20, so a caller of its modules never has to think about this).

    Under capture nobody can wait for the guard's verdict, so a guarded call that leaves the format's domain NaN-fills its output (range bits, heavy
    cancellation) or -- mild cancellation estimate, bit 8 -- keeps an f16f6 result whose error is still ~<= 1e-4 (ops.set_range_check).  This wrapper reads
    the status word the replay left on the device (4 bytes into pinned host memory behind the replay, an event instead of a device-wide synchronise) and,
    when ANY bit is set, runs `fn` again EAGERLY on the same inputs with the whole package in the mode the verdict asks for ('fp32' on bit 16, else
    'bf16x3'), returning that result: what the 'sync' mode of the eager path does, one replay late.

        g = cti_amd.GraphedForward(lambda v, q, a: model(v, q, a), (v, q, a))
        out = g(v, q, a)          # tensors of the captured graph (overwritten by the next call) or, after a trip, of the eager re-run

    fn must be inference code (torch.no_grad is entered here).  `reruns` counts the trips, `last_status` is the last status word."""

    def __init__(self, fn, example_inputs, warmup=2):
        self.fn = fn
        self.static_in = tuple(t.clone() for t in example_inputs)
        self.reruns, self.last_status = 0, 0
        dev = self.static_in[0].device
        s = torch.cuda.Stream(device=dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s), torch.no_grad():
            for _ in range(warmup):                      # eager, on the capture stream: allocator, caches, stream-K workspaces, the guard's status block
                self.fn(*self.static_in)
            torch.cuda.synchronize(dev)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, stream=s):
                self.static_out = self.fn(*self.static_in)
        torch.cuda.current_stream(dev).wait_stream(s)
        self._blk = ops._guard_status_block.get(dev.index if dev.index is not None else torch.cuda.current_device())
        self._host = torch.zeros(1, dtype=torch.int32).pin_memory() if self._blk is not None else None
        self._ev = torch.cuda.Event()

    def __call__(self, *inputs):
        for dst, src in zip(self.static_in, inputs):
            dst.copy_(src, non_blocking=True)
        self.graph.replay()
        if self._blk is None:                            # nothing guarded was captured
            return self.static_out
        self._host.copy_(self._blk[:4].view(torch.int32), non_blocking=True)
        self._ev.record()
        self._ev.synchronize()
        self.last_status = int(self._host[0]) & 0xffffffff
        if not self.last_status:
            return self.static_out
        self.reruns += 1
        old = ops.get_precision()
        ops.set_precision("fp32" if self.last_status & 16 else "bf16x3")
        try:
            with torch.no_grad():
                return self.fn(*self.static_in)
        finally:
            ops.set_precision(old)
