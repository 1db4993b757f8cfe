"""SimGCL (Yu et al., SIGIR'22; reference model/graph/SimGCL.py:12-93), engine-backed.
Config block ``SimGCL: {n_layer, lambda, eps}``; the temperature is the reference's
hard-coded 0.2 (SimGCL.py:48-49)."""
from ._fused import FusedGraphModel


class SimGCL(FusedGraphModel):
    engine_model = "SimGCL"

    def engine_kwargs(self):
        c = self.config['SimGCL']
        return {"n_layers": int(c['n_layer']), "cl_rate": float(c['lambda']), "eps": float(c['eps'])}
