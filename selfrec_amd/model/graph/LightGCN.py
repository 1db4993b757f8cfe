"""LightGCN (He et al., SIGIR'20; reference model/graph/LightGCN.py:10-78), engine-backed.
Config block ``LightGCN: {n_layer}``."""
from ._fused import FusedGraphModel


class LightGCN(FusedGraphModel):
    engine_model = "LightGCN"

    def engine_kwargs(self):
        return {"n_layers": int(self.config['LightGCN']['n_layer'])}

    def should_evaluate(self, epoch):
        return epoch % 5 == 0                     # LightGCN.py:34-35
