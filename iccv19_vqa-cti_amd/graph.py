"""hipGraph capture of a whole training step (reference src/FFOE/trainer.py:97-149,221-269).

The FFOE CTI training step is ~500 kernel launches issued by ~7 ms of host Python for ~7.7 ms of GPU work: the step is host-co-limited.
`GraphedTrainStep` captures forward + loss + backward + FlatAdamaxDP.step() ONCE (torch.cuda.CUDAGraph = hipGraph) and replays it per batch:
the host then issues one hipGraphLaunch per step.  What makes the step capturable is that nothing that changes from step to step is a kernel
ARGUMENT: the learning rate, the completed-step count of Adamax' bias correction and the dropout streams' step counter live in device memory
(cti_adamax_step_g / cti_dropout_g / cti_counter_add), and the library allocates nothing and never synchronises.  Single-rank steps (no
collective) are what is captured and tested; capturing the RCCL all-reduce of a process group aborted the process on this stack (ROCm 7.0 /
torch 2.10, `bench.py --mode train --graph` with CTI_BENCH_FORCE_DIST=1), so multi-rank steps are launched eagerly.  Batches are fed by
copying into static input tensors."""
import torch

from . import ops


class GraphedTrainStep:
    def __init__(self, model, optimizer, loss_fn, example_inputs, example_target, warmup=2):
        """loss_fn(model_output, target) -> scalar tensor.  example_inputs: tuple of tensors (shapes / dtypes are frozen); warmup eager steps run on
        a side stream first (allocator, one-time attributes, weight-norm caches), as torch's capture rules ask.  The warm-up steps DO update the
        parameters."""
        self.model, self.opt, self.loss_fn = model, optimizer, loss_fn
        self.static_in = tuple(t.clone() for t in example_inputs)
        self.static_tgt = example_target.clone()
        self.graph = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(warmup):
                self._step()
            torch.cuda.synchronize()
            ops.invalidate_caches()                      # the captured step must contain the per-step refresh of every derived cache
            with torch.cuda.graph(self.graph, stream=s):
                self.static_loss = self._step()
        torch.cuda.current_stream().wait_stream(s)

    def _step(self):
        self.opt.zero_grad()
        loss = self.loss_fn(self.model(*self.static_in), self.static_tgt)
        loss.backward()
        self.opt.step()
        return loss.detach()

    def __call__(self, inputs, target):
        """One training step on a new batch (copied into the captured tensors); returns the loss tensor of the replay (device, no sync)."""
        for dst, src in zip(self.static_in, inputs):
            dst.copy_(src, non_blocking=True)
        self.static_tgt.copy_(target, non_blocking=True)
        self.graph.replay()
        return self.static_loss
