"""SGL with edge dropout (Wu et al., SIGIR'21; reference model/graph/SGL.py:14-125),
engine-backed.  Config block ``SGL: {n_layer, lambda, drop_rate, aug_type, temp}``."""
from ._fused import FusedGraphModel


class SGL(FusedGraphModel):
    engine_model = "SGL"

    def engine_kwargs(self):
        c = self.config['SGL']
        return {"n_layers": int(c['n_layer']), "cl_rate": float(c['lambda']), "drop_rate": float(c['drop_rate']),
                "aug_type": int(c['aug_type']), "tau": float(c['temp'])}

    def should_evaluate(self, epoch):
        return epoch >= 5                         # SGL.py:45-46
