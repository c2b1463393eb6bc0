"""One training step of the masked-GAT auto-encoder as a HIP graph.

The reference's step (``train.py:109-117``, ``model/train.py:38-46``) is ``loss = model(g)``,
``zero_grad``, ``backward``, ``step``; on the MI355X that is ~180 kernel launches whose gaps cost as
much as a quarter of the step on small graphs.  :class:`CapturedTrainStep` records the whole step
(fresh one-hot node attributes, ``compute_loss`` forward, backward, Adam) once per graph structure
and replays it: every buffer of the recorded step is static, and what changes from step to step
-- the mask permutation, the sampled positive edge ids, the candidate negative pairs, the seeds
of the dropout masks -- is written into fixed device tensors before each replay; no random number
is drawn inside the recorded step.

The arithmetic is the eager step's: the same kernels run in the same order on the same inputs, so a
replay is bitwise equal to the eager step with the same permutation / samples / seeds.
"""
from __future__ import annotations

from typing import Optional

import torch

from .. import _lib
from ..utils.loaddata import _one_hot_float
from ..utils.utils import sample_range


def make_adam(model, lr: float, weight_decay: float, capturable: bool = True):
    """Adam as the reference configures it (``utils/utils.py:24-31``: L2 term added to the gradient),
    as one multi-tensor launch of the library with its step counter on the device (so every instance can be
    recorded into a HIP graph; ``capturable`` is kept for callers written against torch's fused Adam)."""
    from ..optim import FusedAdam
    return FusedAdam(model.parameters(), lr=lr, weight_decay=weight_decay)


def _shared_pool(model):
    """One allocator pool for every recorded step of a model: the steps of one training loop replay
    one after another, never concurrently, and hand tensors (the parameter gradients) to each other."""
    pool = getattr(model, "_magic_graph_pool", None)
    if pool is None:
        pool = model._magic_graph_pool = torch.cuda.graph_pool_handle()
    return pool


def _shared_stream(model):
    """The side stream every step of a model runs on, eager or recorded: autograd keeps a stream per
    gradient accumulator, and a recording whose backward meets accumulators of another stream forks
    that stream into the capture."""
    st = getattr(model, "_magic_step_stream", None)
    if st is None:
        st = model._magic_step_stream = torch.cuda.Stream()
    return st


# Recording must not outlaw what OTHER threads do meanwhile: the process group's watchdog thread polls the events of
# finished collectives (an event query), which the default ("global") capture mode turns into an error that takes the
# process down.  Calls of the recording thread itself are still checked.
_CAPTURE = {"capture_error_mode": "thread_local"}


class CapturedTrainStep:
    """``step = CapturedTrainStep(model, optimizer, g); loss = step()`` -- one training step on the
    fixed graph ``g`` per call.

    ``g`` needs ``ndata['type']`` (the one-hot node attributes are rebuilt inside the step, as the
    reference's loaders rebuild them) and one-hot ``edata['attr']``.  ``loss_scale`` multiplies the loss
    before backward (``train.py:113``: ``loss /= n_train``).  ``grad_exchange`` (a
    :class:`magic_amd.parallel.FlatGradAllReduce`) runs between backward and the optimizer step; the
    step is then recorded as two graphs around that collective.

    ``capture=False`` (the default) runs the same statements eagerly on the caller's stream.  Measured
    on MI355X / ROCm 7.2 the replayed step is not faster than the eager one at any of the benchmark's
    shapes (its ~180 launches keep the device busy back to back either way; what separates them is
    the device-side kernel boundary, which a replay does not remove), so recording is an option for
    callers whose host thread is busy elsewhere, not the default.  With ``capture=True`` the
    ``optimizer`` must be capturable (:func:`make_adam`) and every step of the model runs on one side
    stream, eager warm-up calls included."""

    def __init__(self, model, optimizer, g, n_dim: int, loss_scale: float = 1.0, grad_exchange=None,
                 warmup: int = 2, capture: bool = False, capture_optimizer: bool = True):
        self.model, self.opt, self.g, self.n_dim = model, optimizer, g, n_dim
        self.loss_scale, self.exchange = float(loss_scale), grad_exchange
        dev = g.device
        n, e = g.num_nodes(), g.num_edges()
        self.n, self.e = n, e
        self.thr = min(10000, n)
        if e < self.thr:
            raise ValueError("Sample larger than population or is negative")   # random.sample's own error
        self.ntype = g.ndata["type"].view(-1).long()
        g.structure.ensure_csr()
        # static inputs of the recorded step
        self.perm = torch.zeros(n, dtype=torch.int64, device=dev)
        self.pos = torch.zeros(self.thr, dtype=torch.int64, device=dev)
        self.n_masks = len(model.encoder.gats) + 1
        self.seeds = torch.zeros(self.n_masks, dtype=torch.int64, device=dev)
        from .autoencoder import negative_candidate_count
        self.n_neg = negative_candidate_count(self.thr)
        self.neg_u = torch.zeros(self.n_neg, dtype=torch.int64, device=dev)   # candidate negative pairs
        self.neg_v = torch.zeros(self.n_neg, dtype=torch.int64, device=dev)
        self.neg_ok = torch.zeros(self.n_neg, dtype=torch.bool, device=dev)   # ... and which of them count
        self.loss = torch.zeros((), dtype=torch.float32, device=dev)
        self._graphs = None
        self._capture = capture
        self._capture_opt = capture_optimizer
        self._warmup = warmup
        self._calls = 0

    # ------------------------------------------------------------------ pieces of one step
    def _forward_backward(self, inputs=None):
        """``inputs``: keyword overrides of ``compute_loss`` (eager mode), or None for the static
        input buffers of the recorded step."""
        g, m = self.g, self.model
        g.ndata["attr"] = _one_hot_float(self.ntype, self.n_dim)
        g.mark_node_attr_onehot()
        if inputs is None:
            inputs = dict(perm=self.perm, pos_idx=self.pos, neg_pairs=(self.neg_u, self.neg_v),
                          neg_valid=self.neg_ok, drop_seeds=self.seeds)
        loss = m.compute_loss(g, **inputs)
        if self.loss_scale != 1.0:
            loss = loss * self.loss_scale
        if self.exchange is not None:
            self.exchange.zero()
        else:
            self.opt.zero_grad(set_to_none=True)
        loss.backward()
        self.loss.copy_(loss.detach())
        # the masked attribute tensor carries autograd history (the token was written into it in
        # place): drop it now instead of keeping that graph alive until the next step
        g.ndata.pop("attr", None)

    def _eager(self, inputs=None):
        self._forward_backward(inputs)
        if self.exchange is not None:
            self.exchange.allreduce()
        self.opt.step()

    def _record(self):
        """Record the step (recording launches nothing; the caller replays it afterwards)."""
        ws_before = {k: v.data_ptr() for k, v in _lib._ws_cache.items()}
        # gradients left by an earlier step may live in another recorded step's memory pool: release
        # them here, not inside the recording
        for p_ in self.model.parameters():
            p_.grad = None
        torch.cuda.synchronize()
        if self.exchange is None:
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, pool=_shared_pool(self.model), stream=_shared_stream(self.model), **_CAPTURE):
                self._forward_backward()
                if self._capture_opt:
                    self.opt.step()
            graphs = [gr]
        else:
            ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with torch.cuda.graph(ga, pool=_shared_pool(self.model), stream=_shared_stream(self.model), **_CAPTURE):
                self._forward_backward()
            with torch.cuda.graph(gb, pool=_shared_pool(self.model), stream=_shared_stream(self.model), **_CAPTURE):
                self.opt.step()
            graphs = [ga, gb]
        if {k: v.data_ptr() for k, v in _lib._ws_cache.items()} != ws_before:
            raise _lib.MagicHipError("a kernel workspace grew while the step was being recorded")
        # the recorded launches hold raw pointers into the kernel workspaces: keep those buffers alive
        # even if a later, larger call makes the library replace them
        self._keep = list(_lib._ws_cache.values())
        self._graphs = graphs

    # ------------------------------------------------------------------ per-step inputs
    def _load(self, name, dst, value):
        """Copy a tensor into a static device tensor; host tensors go through pinned memory from
        torch's caching host allocator (which does not recycle a block before its copy has run)."""
        if value.device.type != "cuda":
            pin = torch.empty(value.shape, dtype=dst.dtype, pin_memory=True)
            pin.copy_(value)
            value = pin
        dst.copy_(value, non_blocking=True)

    def set_inputs(self, perm=None, pos_idx=None, seeds=None, neg_pairs=None, neg_valid=None):
        """Defaults follow the reference: ``torch.randperm`` from torch's CPU generator (the device
        generator with ``mask_rng='device'``), ``random.sample(range(E), T)``, T uniformly drawn
        candidate pairs of which the non-edges that are no self-loops count (the stand-in for DGL's
        negative sampler, drawn from torch's device generator), dropout seeds from torch's CPU
        generator.  ``neg_pairs`` may hold up to ``negative_candidate_count(T)`` pairs, T = min(10000, N)."""
        from .autoencoder import _negative_candidates
        dev = self.g.device
        if perm is None:
            perm = torch.randperm(self.n, device=dev) if self.model.mask_rng == "device" else torch.randperm(self.n)
        self._load("perm", self.perm, torch.as_tensor(perm).long())
        if pos_idx is None:
            pos_idx = sample_range(self.e, self.thr)
        self._load("pos", self.pos, torch.as_tensor(pos_idx).long())
        if seeds is None:
            seeds = torch.randint(0, 2 ** 62, (self.n_masks,))
        self._load("seeds", self.seeds, torch.as_tensor(seeds).long())
        if neg_pairs is None:
            u, v, ok = _negative_candidates(self.g, self.thr)
        else:
            u, v = (torch.as_tensor(t).long().to(dev) for t in neg_pairs)
            ok = torch.ones_like(u, dtype=torch.bool) if neg_valid is None else torch.as_tensor(neg_valid).to(dev)
            if u.numel() > self.n_neg:
                raise ValueError(f"at most {self.n_neg} candidate negative pairs fit the recorded step")
        k = u.numel()                       # fewer injected pairs than slots: the rest does not count
        self.neg_ok.zero_()
        self.neg_u[:k].copy_(u)
        self.neg_v[:k].copy_(v)
        self.neg_ok[:k].copy_(ok)

    def __call__(self, **inputs) -> torch.Tensor:
        """Run exactly one training step; returns the (scaled) loss as a device scalar that the next
        call overwrites.  The first ``warmup`` calls run eagerly (they size the allocator pools and
        the kernel workspaces and create the optimizer state); the next call records the step and
        every call from then on replays it."""
        self.model.train()
        if not self._capture:
            # plain eager step on the caller's stream: the inputs go to compute_loss as they are
            # (missing ones are drawn there as the reference draws them)
            kw = {k: v for k, v in inputs.items() if v is not None}
            if "seeds" in kw:
                kw["drop_seeds"] = torch.as_tensor(kw.pop("seeds")).long().to(self.g.device)
            if "perm" in kw:
                kw["perm"] = torch.as_tensor(kw["perm"]).to(self.g.device)
            self._calls += 1
            self._eager(kw)
            return self.loss
        self.set_inputs(**inputs)
        cur, st = torch.cuda.current_stream(), _shared_stream(self.model)
        st.wait_stream(cur)                 # the inputs were written on the caller's stream
        with torch.cuda.stream(st):
            self._step()
        cur.wait_stream(st)
        return self.loss

    def _step(self):
        if self._calls < self._warmup:
            self._calls += 1
            self._eager()
            return
        if self._graphs is None:
            self._record()
        self._calls += 1
        self._graphs[0].replay()
        if self.exchange is not None:
            self.exchange.allreduce()
            self._graphs[1].replay()
        elif not self._capture_opt:
            self.opt.step()
