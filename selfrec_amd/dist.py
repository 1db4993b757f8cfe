"""(e) Multi-GPU: the fused step with the graph and the tables row-sharded over the ranks of one node.

One process per GPU, ``torch.distributed`` (backend "nccl" = RCCL over xGMI).  The partition SURVEY.md
8(e) describes -- graph rows = embedding rows -- is a LAYOUT of ``engine.FusedTrainer``, not a second
engine: nodes are dealt round-robin (node p -> rank p % G, local row p // G: power-law rows balance
without a partitioner), every (.., d) table is kept in all-gather order, each rank owns one slice of
the parameters, the Adam moments and every layer output, and computes it with the same kernels from
its CSR rows:

    forward layer k    Y_k[own] = A[own, :] . Y_(k-1)           then all-gather Y_k
    backward layer k   H_k[own] = A[own, :] . H_(k+1) + ...      then all-gather H_k   (A is symmetric:
                       the transpose product is the same local row product, so there is no
                       reduce-scatter -- 2L all-gathers of N.d.4 bytes per step, plus one for E0)
    batch level        the sampler is replicated (same seed, same MT19937 stream => identical batches on
                       every rank); with whole tables on every rank the O(batch) losses and their
                       gradients are simply recomputed everywhere -- no collective, and the step keeps
                       its device-side cursor, so it is captured in a hipGraph like the 1-GPU step
    optimiser          Adam on the owned rows only
    graph              each rank normalises its own CSR rows on the device (degrees of its rows, one all-gather of
                       the D^-1/2 vector, then the values: data/device_graph.ShardedDeviceGraph), which is also
                       how SGL's edge-dropped views are rebuilt every epoch

On the xGMI mesh an all-gather of (N/G).d.4-byte slices moves each slice over its own link; at the
Yelp2018 shape (17.8 MB tables) the step is latency-bound and does not beat one GPU -- the layout is
for graphs whose tables do not fit or whose SpMM dominates (the 1 M x 500 k configuration).
"""
from __future__ import annotations

from .engine import FusedTrainer, shard_adjacency  # noqa: F401  (shard_adjacency: public helper)


class ShardedTrainer(FusedTrainer):
    """``FusedTrainer`` over the default process group (all five models).  Same constructor, same
    ``begin_epoch / step / read_losses / embeddings``; every rank must be seeded identically."""

    def __init__(self, data, emb_size, **kw):
        kw.pop("backend", None)
        super().__init__(data, emb_size, shard=True, **kw)

    def parameters_full(self):
        return self.user_emb, self.item_emb
