"""Matrix factorisation + BPR (reference model/graph/MF.py:8-61), engine-backed."""
from ._fused import FusedGraphModel


class MF(FusedGraphModel):
    engine_model = "MF"

    def should_evaluate(self, epoch):
        return epoch % 5 == 0                     # MF.py:30-31
