"""Common body of the engine-backed LightGCN-family recommenders.

Each concrete class (MF, LightGCN, XSimGCL, SimGCL, SGL in this package) reads the same
``conf`` keys as its reference namesake and exposes the same lifecycle
(``train / save / predict``, ``user_emb / item_emb / best_user_emb / best_item_emb``), but its
``train()`` drives ``selfrec_amd.engine.FusedTrainer`` instead of a python loop of ATen ops.
"""
import random

import torch

from ...base.graph_recommender import GraphRecommender
from ...engine import EpochPrefetcher, FusedTrainer


def _as_bool(v):
    """YAML gives real booleans, but a quoted 'false' / 'off' / '0' must not read as True."""
    if isinstance(v, str):
        return v.strip().lower() not in ("", "0", "false", "no", "off", "none")
    return bool(v)


class FusedGraphModel(GraphRecommender):
    engine_model = None            # "MF" | "LightGCN" | ...

    def engine_kwargs(self):
        return {}

    def should_evaluate(self, epoch):
        return True

    def __init__(self, conf, training_set, test_set, **kwargs):
        super().__init__(conf, training_set, test_set, **kwargs)
        get = getattr(self.config, 'get', lambda k, d=None: d)
        # (any embedding.size up to 256: the engine stores the tables
        # zero-padded to the next width its kernels serve -- engine.FusedTrainer.d_valid; wider rows raise there)
        # the in-kernel perturbation noise follows torch's seed (torch.manual_seed / `seed` in the conf), like the
        # reference's torch.rand_like does
        seed = get('seed', None)
        rng_seed = (int(seed) if seed is not None else torch.initial_seed()) & ((1 << 63) - 1)
        # "f32" | "split" | absent: the arithmetic of InfoNCE's two products belongs to THIS model's trainer and travels with
        # every loss call (no process-wide state: a model neither inherits nor leaves behind another model's setting).  Absent
        # = the library default: the reference's fp32 products (loss_torch.py:46-47); "split" is the faster opt-in.
        prec = get('engine.nce_precision', None)
        self.nce_precision = None if prec is None else str(prec)
        self.trainer = FusedTrainer(self.data, self.emb_size, model=self.engine_model, lr=self.lRate,
                                    reg=self.reg, batch_size=self.batch_size, rng_seed=rng_seed,
                                    use_graph=_as_bool(get('engine.hipgraph', True)), nce_precision=self.nce_precision,
                                    **self.engine_kwargs())
        self.exact_sampling = _as_bool(get('sampler.python_state', True))

    def train(self):
        tr = self.trainer
        tr.set_nce_precision(self.nce_precision)
        if self.exact_sampling:           # consume the global `random` stream like the reference
            tr.seed_sampler_from_python()
        else:
            tr.sampler.seed(random.getrandbits(63))
        prefetch = EpochPrefetcher(tr)
        prefetch.start()
        for epoch in range(self.maxEpoch):
            tr.upload_epoch(prefetch.take())
            evaluate_now = self.should_evaluate(epoch)
            if epoch + 1 < self.maxEpoch:
                prefetch.start()          # host samples epoch e+1 while the device runs epoch e
            for n in range(tr.epoch_batches):
                tr.step()
                if n % 100 == 0 and n > 0:
                    bpr, reg, cl = tr.read_losses()
                    print('training:', epoch + 1, 'batch', n, 'rec_loss:', bpr, 'cl_loss', cl)
            self.user_emb, self.item_emb = tr.embeddings()
            if evaluate_now:
                self.fast_evaluation(epoch)
        if self.exact_sampling:
            tr.sampler.push_state_to_python()
        if hasattr(self, 'best_user_emb'):
            self.user_emb, self.item_emb = self.best_user_emb, self.best_item_emb

    def save(self):
        ue, ie = self.trainer.embeddings()
        self.best_user_emb, self.best_item_emb = ue.clone(), ie.clone()

    def predict(self, u):
        uid = self.data.get_user_id(u)
        with torch.no_grad():
            return torch.matmul(self.user_emb[uid], self.item_emb.transpose(0, 1)).cpu().numpy()


def fused_train_of_reference_class(self):
    """``train()`` of an UNMODIFIED reference model class (``dropin.install(fuse=True)``; dropin.maybe_fuse checked the
    file's SHA-256): the loop of model/graph/{MF,LightGCN,XSimGCL,SimGCL,SGL}.py -- sample, forward, losses, backward,
    Adam, per-epoch ``model()`` + ``fast_evaluation`` -- on ``engine.FusedTrainer`` instead of ~190 eager launches per
    step.  What the file's own code still does: ``__init__`` (so the tables are ITS xavier draws under torch's seed),
    ``save()`` / ``predict()`` (they call the file's torch encoder, whose parameters alias the engine's table).
    A patched ``torch.rand_like`` (parity harnesses inject the reference's CPU noise that way) is honoured: the
    perturbation then takes that stream instead of the in-kernel counter RNG (and the step is launched eagerly)."""
    import importlib
    import torch as _torch
    name = type(self).__name__
    adapter = getattr(importlib.import_module(f"{__package__}.{name}"), name)       # config keys -> engine arguments
    kw = adapter.engine_kwargs(self)
    model = self.model.cuda()
    params = model.embedding_dict
    get = getattr(self.config, 'get', lambda k, d=None: d)
    seed = get('seed', None)
    rng_seed = (int(seed) if seed is not None else _torch.initial_seed()) & ((1 << 63) - 1)
    patched = not isinstance(_torch.rand_like, type(_torch.empty))       # (a python function where torch has a builtin)
    noise_fn = (lambda shape: _torch.rand_like(_torch.empty(shape))) if patched and name in ("XSimGCL", "SimGCL") else None
    tr = self.trainer = FusedTrainer(self.data, self.emb_size, model=name, lr=self.lRate, reg=self.reg,
                                     batch_size=self.batch_size, rng_seed=rng_seed, noise_fn=noise_fn,
                                     use_graph=_as_bool(get('engine.hipgraph', True)),
                                     user_emb=params['user_emb'].detach().cpu(), item_emb=params['item_emb'].detach().cpu(),
                                     **kw)
    # the file's encoder keeps working on the trained values: its parameters become views of the engine's table
    params['user_emb'].data, params['item_emb'].data = tr.user_emb, tr.item_emb
    tr.seed_sampler_from_python()                        # the global `random` stream, as next_batch_pairwise consumes it
    limit = getattr(self, "_fused_step_limit", None)     # (parity tooling: stop after this many steps of epoch 1)
    prefetch = EpochPrefetcher(tr)
    prefetch.start()
    for epoch in range(self.maxEpoch):
        tr.upload_epoch(prefetch.take())
        if epoch + 1 < self.maxEpoch and limit is None:
            prefetch.start()
        for n in range(tr.epoch_batches if limit is None else min(limit, tr.epoch_batches)):
            tr.step()
            if n % 100 == 0 and n > 0:
                bpr, reg, cl = tr.read_losses()
                if name in ("MF", "LightGCN"):
                    print('training:', epoch + 1, 'batch', n, 'batch_loss:', bpr + reg)
                else:
                    print('training:', epoch + 1, 'batch', n, 'rec_loss:', bpr, 'cl_loss', cl)
        if limit is not None:
            break
        self.user_emb, self.item_emb = tr.embeddings()   # (= `with torch.no_grad(): self.model()` of the file)
        if adapter.should_evaluate(self, epoch):
            self.fast_evaluation(epoch)
    tr.sampler.push_state_to_python()
    if limit is None:
        self.user_emb, self.item_emb = self.best_user_emb, self.best_item_emb      # (as the file: raises if never saved)
