"""Common body of the engine-backed LightGCN-family recommenders.

Each concrete class (MF, LightGCN, XSimGCL, SimGCL, SGL in this package) reads the same
``conf`` keys as its reference namesake and exposes the same lifecycle
(``train / save / predict``, ``user_emb / item_emb / best_user_emb / best_item_emb``), but its
``train()`` drives ``selfrec_amd.engine.FusedTrainer`` instead of a python loop of ATen ops.
"""
import random

import torch

from ..._lib import SelfrecHipError
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
        if self.engine_model in ("XSimGCL", "SimGCL", "SGL") and int(self.emb_size) not in (64, 128):
            raise SelfrecHipError(f"{self.engine_model}: embedding.size = {self.emb_size} -- the fused InfoNCE kernels serve "
                                  f"64 and 128 (use the op-level drop-in tier, selfrec_amd.dropin, for other sizes)")
        # the in-kernel perturbation noise follows torch's seed (torch.manual_seed / `seed` in the conf), like the
        # reference's torch.rand_like does
        seed = get('seed', None)
        rng_seed = (int(seed) if seed is not None else torch.initial_seed()) & ((1 << 63) - 1)
        self.trainer = FusedTrainer(self.data, self.emb_size, model=self.engine_model, lr=self.lRate,
                                    reg=self.reg, batch_size=self.batch_size, rng_seed=rng_seed,
                                    use_graph=_as_bool(get('engine.hipgraph', True)), **self.engine_kwargs())
        self.exact_sampling = _as_bool(get('sampler.python_state', True))
        precision = get('engine.nce_precision', None)          # "f32" | "bf16x3" (process-wide; default bf16x3)
        if precision is not None:
            from ... import ops
            ops.set_infonce_precision(str(precision))

    def train(self):
        tr = self.trainer
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
