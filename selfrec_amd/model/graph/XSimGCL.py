"""XSimGCL (Yu et al., TKDE'23; reference model/graph/XSimGCL.py:12-101), engine-backed.
Config block ``XSimGCL: {n_layer, l_star, lambda, eps, tau}``."""
from ._fused import FusedGraphModel


class XSimGCL(FusedGraphModel):
    engine_model = "XSimGCL"

    def engine_kwargs(self):
        c = self.config['XSimGCL']
        return {"n_layers": int(c['n_layer']), "layer_cl": int(c['l_star']), "cl_rate": float(c['lambda']),
                "eps": float(c['eps']), "tau": float(c['tau'])}
