import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_ops():
    import numpy as np
    return np.load(os.path.join(GOLDEN, "ops_sampler_graph.npz"))


@pytest.fixture(scope="session")
def golden_models():
    import numpy as np
    return np.load(os.path.join(GOLDEN, "models.npz"))


@pytest.fixture(scope="session")
def golden_meta():
    import json
    with open(os.path.join(GOLDEN, "meta.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden_edges():
    """reference runs of the configurations make_golden.py's EDGE_CASES lists (d = 256, one / four layers, l* = 0 / L)"""
    import json
    import numpy as np
    with open(os.path.join(GOLDEN, "edges_meta.json")) as f:
        meta = json.load(f)
    return np.load(os.path.join(GOLDEN, "edges.npz")), meta


EDGE_TAGS = ["XSimGCL_d256", "LightGCN_d256", "XSimGCL_L1_s0", "XSimGCL_L1_s1", "XSimGCL_L4_s0", "XSimGCL_L4_s4",
             "LightGCN_L1", "LightGCN_L4", "SimGCL_L1", "SimGCL_L4", "SGL_L1", "SGL_L4"]


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no HIP device in this container (run on the GPU box with -m gpu)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def _tiny_interaction(golden_ops):
    from selfrec_amd import synth
    from selfrec_amd.data.ui_graph import Interaction
    g = golden_ops
    gm = __import__("numpy").load(os.path.join(GOLDEN, "models.npz"))
    train = synth.as_triples(g["graph_train_u_raw"], g["graph_train_i_raw"])
    test = synth.as_triples(gm["test_u_ids_raw"], gm["test_i_ids_raw"])
    return Interaction({}, train, test)


@pytest.fixture(scope="session")
def tiny_data(golden_ops):
    """Interaction object over the golden 200 x 300 graph (names = raw ids as strings).
    Shared: tests must not mutate it (use fresh_tiny_data with the drop-in sampler)."""
    return _tiny_interaction(golden_ops)


@pytest.fixture()
def fresh_tiny_data(golden_ops):
    return _tiny_interaction(golden_ops)
