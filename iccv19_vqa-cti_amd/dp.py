"""Data-parallel training step for models built on the CTI modules.

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).  All trainable parameters, their gradients and the
Adamax state live in FOUR flat fp32 buffers; `param.data` is a view into the first, so
  * backward leaves each gradient in the tensor autograd produced (zero_grad() sets `.grad = None`, so AccumulateGrad keeps that tensor
    instead of launching one add per parameter -- 344 of them in the FFOE CTI model) and ONE gather kernel packs them into the flat
    gradient buffer (the reference's _get_flat_grads torch.cat + _set_flat_grads copies, src/FFOE/trainer.py:245-263); after step()
    `param.grad` is a view of that buffer (the averaged, unclipped gradient),
  * ONE all-reduce(sum) of that buffer is the only communication of the step,
  * scale by 1/(world*update_freq), global-norm clip and the Adamax update are two HIP kernels with no host sync
    (the reference blocks on grad_norm.item(), src/utils.py:324).
Semantics restated from the reference: loss is divided by the LOCAL batch (trainer.py:189-190), gradients are summed over ranks
and divided by world_size * update_freq, clip coefficient max_norm / (norm + 1e-6) applied only when < 1 (utils.py:323-328),
Adamax with betas (0.9, 0.999), eps 1e-8, no weight decay (torch.optim.Adamax defaults, train.py:34)."""
import ctypes

import torch
import torch.distributed as dist

from . import _lib as L
from . import ops


class FlatAdamaxDP:
    ALIGN = 64                                                  # floats: 256-B parameter alignment inside the flat buffers

    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, clip_norm=0.25, update_freq=1, process_group=None, force_collective=False):
        """force_collective: issue the broadcast / all-reduce even in a one-rank group (exercises RCCL on a 1-GPU box; identity there)."""
        self.force_collective = bool(force_collective)
        self.params = [p for p in model.parameters() if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev = self.params[0].device
        if dev.type != "cuda":
            raise L.CtiError("FlatAdamaxDP runs on the GPU only (HIP update kernels, RCCL all-reduce)")
        for p in self.params:
            if p.dtype != torch.float32 or p.device != dev:
                raise TypeError("all trainable parameters must be fp32 on one device")
        # every parameter starts on a 256-B boundary of the flat buffers: the kernels' 16-B load paths (plane splits, MFMA fragment
        # loads) test the pointer alignment of their operands, and a packed layout would send most weights down the scalar paths.
        # The padding stays zero in all four buffers (zero gradient -> zero Adamax update), so norms and updates are unaffected.
        self.n_params = sum(p.numel() for p in self.params)
        self.n = sum((p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN for p in self.params)
        self.flat_p = torch.zeros(self.n, device=dev, dtype=torch.float32)
        self.flat_g = torch.zeros(self.n, device=dev, dtype=torch.float32)
        self.exp_avg = torch.zeros(self.n, device=dev, dtype=torch.float32)
        self.exp_inf = torch.zeros(self.n, device=dev, dtype=torch.float32)
        off = 0
        self.offsets = []
        for p in self.params:                                   # named_parameters() order, like Trainer._get_flat_grads
            k = p.numel()
            self.flat_p[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.flat_p[off:off + k].view(p.shape)
            p.grad = None
            self.offsets.append(off)
            off += (k + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self._table = (ctypes.c_int64 * (3 * len(self.params)))()
        for i, (p, o) in enumerate(zip(self.params, self.offsets)):
            self._table[3 * i + 1], self._table[3 * i + 2] = o, p.numel()
        self._grad_views = [self.flat_g[o:o + p.numel()].view(p.shape) for p, o in zip(self.params, self.offsets)]   # built once, re-attached every step
        lib = L.lib()
        self.partial = torch.empty(lib.cti_optim_workspace_bytes() // 4, device=dev, dtype=torch.float32)
        self.grad_norm = torch.zeros(1, device=dev, dtype=torch.float32)
        # what changes from step to step lives in device memory (learning rate, completed-step count, the dropout streams' step counter), so a
        # whole training step -- forward, backward, this step() -- captured in a hipGraph replays correctly (tools/graph_train.py)
        self.lr_dev = torch.full((1,), float(lr), device=dev, dtype=torch.float32)
        self.step_dev = torch.zeros(1, device=dev, dtype=torch.int64)
        ops._rng_tensor(dev)
        self._lr = float(lr)
        self.betas, self.eps, self.clip_norm, self.update_freq = betas, eps, clip_norm, update_freq
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        self.step_count = 0

    @property
    def lr(self):
        return self._lr

    @lr.setter
    def lr(self, value):
        """Learning-rate schedules assign here between steps (src/FFOE/train.py:75-83): the device copy follows (not while capturing)."""
        self._lr = float(value)
        self.lr_dev.fill_(self._lr)

    def broadcast_parameters(self, src=0):
        """Identical initial parameters on every rank: one broadcast of the flat buffer."""
        if self.needs_collective():
            dist.broadcast(self.flat_p, src=src, group=self.pg)
            ops.invalidate_caches()                              # flat_p was overwritten behind autograd's version counters

    def zero_grad(self, set_to_none=True):
        """set_to_none=True (default, like torch.optim): no fill kernel -- autograd then keeps the gradient tensors it produces and
        cti_flat_gather zeroes whatever has none.  False: zero the flat buffer and leave every .grad a view of it (backward then accumulates
        in place, one add per parameter)."""
        if set_to_none:
            for p in self.params:
                p.grad = None
        else:
            self.flat_g.zero_()
            for p, gv in zip(self.params, self._grad_views):
                p.grad = gv

    def gather_grads(self):
        """param.grad tensors -> flat_g (one kernel); afterwards every param.grad is a view of its flat_g slot.  The host table (slot starts
        and counts filled once) is reused: its entries ride in the kernel arguments of this call."""
        tbl = self._table
        keep = []
        f32, dev = torch.float32, self.flat_g.device
        i = 0
        for p in self.params:
            g = p.grad
            if g is None:
                tbl[i] = 0
            else:
                if g.dtype is not f32 or g.device != dev or g.shape != p.shape:
                    raise TypeError("gradient of a %s parameter is %s %s on %s" % (tuple(p.shape), g.dtype, tuple(g.shape), g.device))
                if not g.is_contiguous():
                    g = g.contiguous()
                    keep.append(g)
                tbl[i] = g.data_ptr()
            i += 3
        L.check(L.lib().cti_flat_gather(tbl, len(self.params), self.flat_g.data_ptr(), self.n, ops._stream()), "cti_flat_gather")
        for p, gv in zip(self.params, self._grad_views):         # drops the gathered tensors: stream-ordered, the allocator reuses them after the gather
            p.grad = gv
        del keep

    def needs_collective(self):
        """True when step() issues the RCCL all-reduce (more than one rank, or force_collective in an initialised one-rank group)."""
        return self.world > 1 or (self.force_collective and dist.is_available() and dist.is_initialized())

    def all_reduce_grads(self):
        """The ONE collective of the step: all-reduce(sum) of the flat gradient buffer over xGMI (no-op without a group)."""
        if self.needs_collective():
            dist.all_reduce(self.flat_g, op=dist.ReduceOp.SUM, group=self.pg)

    def apply_update(self):
        """scale by 1/(world*update_freq) + global norm -> clip + Adamax -> counters: three kernels on device-resident scalars, no host sync,
        capturable (the second half of GraphedTrainStep's split form)."""
        self.step_count += 1                                     # host mirror (eager); the kernels read the device counter
        st = ops._stream()
        lib = L.lib()
        L.check(lib.cti_flat_scale_sumsq(self.flat_g.data_ptr(), self.n, 1.0 / (self.world * self.update_freq), self.partial.data_ptr(), st),
                "cti_flat_scale_sumsq")
        L.check(lib.cti_adamax_step_g(self.flat_p.data_ptr(), self.flat_g.data_ptr(), self.exp_avg.data_ptr(), self.exp_inf.data_ptr(), self.n,
                                      self.partial.data_ptr(), float(self.clip_norm), self.lr_dev.data_ptr(), self.betas[0], self.betas[1], self.eps,
                                      self.step_dev.data_ptr(), self.grad_norm.data_ptr(), st), "cti_adamax_step_g")
        L.check(lib.cti_counter_add(self.step_dev.data_ptr(), 1, st), "cti_counter_add")
        ops.rng_advance(self.flat_p.device)                      # the next step's dropout masks differ, also when this step is a graph replay
        ops.invalidate_caches()                                  # the kernel wrote the parameters behind autograd's version counters
        return self.grad_norm

    def step(self):
        """Call after backward() of the last micro-batch.  Returns the device tensor holding the pre-clip gradient norm.
        = gather_grads() -> all_reduce_grads() -> apply_update(); GraphedTrainStep captures the first and the last into two hipGraphs and
        issues the collective between them."""
        self.gather_grads()
        self.all_reduce_grads()
        return self.apply_update()

    def steps_done(self):
        """Completed steps as the device counts them (= step_count in eager use; graph replays advance only the device counter).  Synchronises."""
        self.step_count = int(self.step_dev.item())
        return self.step_count

    def state_dict(self):
        """The `torch.optim.Adamax.state_dict()` format the reference saves as `optimizer_state` (src/utils.py:104) and reloads with
        `optim.load_state_dict` (src/FFOE/main.py:127, trainer.py:87): per-parameter `state[i] = {step, exp_avg, exp_inf}` in the order of
        `filter(requires_grad, model.parameters())` (src/FFOE/train.py:34) plus one param group.  Tensors are CLONES (later steps do not
        mutate a dict the caller holds); parameters that have not been stepped yet have no entry, like torch."""
        state = {}
        self.steps_done()
        if self.step_count > 0:
            for i, (p, o) in enumerate(zip(self.params, self.offsets)):
                k = p.numel()
                state[i] = {"step": torch.tensor(float(self.step_count)),
                            "exp_avg": self.exp_avg[o:o + k].view(p.shape).clone(),
                            "exp_inf": self.exp_inf[o:o + k].view(p.shape).clone()}
        group = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": 0, "foreach": None, "maximize": False,
                 "differentiable": False, "capturable": False, "params": list(range(len(self.params)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        """Accepts what state_dict() returns, i.e. a torch.optim.Adamax state dict over the same parameter list (torch 1.1's integer `step`
        and torch 2.x's tensor `step` both).  Raises on any layout mismatch (parameter count, shapes, unequal step counts, weight decay)
        instead of mis-assigning moments."""
        if "param_groups" not in sd or "state" not in sd:
            raise ValueError("expected a torch.optim.Adamax state_dict ({'state', 'param_groups'})")
        if len(sd["param_groups"]) != 1:
            raise ValueError("FlatAdamaxDP holds one parameter group, the state dict has %d" % len(sd["param_groups"]))
        grp = sd["param_groups"][0]
        ids = list(grp["params"])
        if len(ids) != len(self.params):
            raise ValueError("the state dict covers %d parameters, the model has %d trainable ones" % (len(ids), len(self.params)))
        if float(grp.get("weight_decay", 0)) != 0.0:
            raise ValueError("weight_decay != 0 is not supported (the reference trains with torch.optim.Adamax defaults)")
        state = sd["state"]
        steps = set()
        for i, pid in enumerate(ids):
            st = state.get(pid, state.get(str(pid)))
            if st is None:
                continue
            for key in ("exp_avg", "exp_inf"):
                if tuple(st[key].shape) != tuple(self.params[i].shape):
                    raise ValueError("parameter %d: %s has shape %s, expected %s" % (i, key, tuple(st[key].shape), tuple(self.params[i].shape)))
            steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError("per-parameter step counts differ (%s): the fused kernel keeps one bias-correction step" % sorted(steps))
        # a parameter torch never stepped (no gradient so far) has no entry: zero moments here, which is what torch starts it from
        self.exp_avg.zero_(); self.exp_inf.zero_()
        for i, pid in enumerate(ids):
            st = state.get(pid, state.get(str(pid)))
            if st is None:
                continue
            o, k = self.offsets[i], self.params[i].numel()
            self.exp_avg[o:o + k].copy_(st["exp_avg"].reshape(-1))
            self.exp_inf[o:o + k].copy_(st["exp_inf"].reshape(-1))
        self.step_count = steps.pop() if steps else 0
        self.step_dev.fill_(self.step_count)
        self.lr = float(grp.get("lr", self.lr))
        self.betas = tuple(grp.get("betas", self.betas))
        self.eps = float(grp.get("eps", self.eps))
        ops.invalidate_caches()
