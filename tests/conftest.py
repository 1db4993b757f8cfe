import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "selfcheck: compares this library with itself (A/B), not with the oracle: collected last")


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


# Collection order of the GPU suite (the driver runs `pytest -x`: the first failure hides everything behind it).  Parity
# against the oracle / the reference's own runs comes FIRST -- every BASELINE.json shape, then every kernel, then the engine
# and the drop-in tiers -- and tests that compare this library with ITSELF (hipGraph replay vs eager launches, staged vs
# in-line epochs, fused vs separate Adam: marked `selfcheck`) come last, so that an A/B disagreement cannot keep a parity
# test from running.
GPU_FILE_ORDER = ["test_gpu_shapes.py", "test_gpu_kernels.py", "test_gpu_engine.py", "test_gpu_dropin.py", "test_gpu_f4.py",
                  "test_gpu_cols.py", "test_gpu_multiproc.py", "test_gpu_selfcheck.py"]


def _gpu_rank(item):
    name = os.path.basename(str(item.fspath))
    where = GPU_FILE_ORDER.index(name) if name in GPU_FILE_ORDER else len(GPU_FILE_ORDER)
    return (1 if "selfcheck" in item.keywords else 0, where)


def pytest_collection_modifyitems(config, items):
    gpu = [it for it in items if "gpu" in it.keywords]
    if gpu:
        ordered = iter(sorted(gpu, key=_gpu_rank))          # stable: the order inside a file stays
        items[:] = [next(ordered) if "gpu" in it.keywords else it for it in items]
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


def host_batch_segments(u, i, j, pad, user_row0=0, item_row0=0):
    """One batch's row groups and (slot, role) lists as include/selfrec_hip.h defines them (srh_batch_segments_t), built
    independently of csrc/sampler.cpp with numpy: dict of int32 arrays (seg_rows, seg_end, seg, seg_a: 3 `pad` entries;
    seg_b: `pad`; the three counts)."""
    import numpy as np
    u, i, j = (np.asarray(a, dtype=np.int64) for a in (u, i, j))
    uu, ui = np.unique(u), np.unique(i)
    un = np.setdiff1d(np.unique(j), ui)
    slot = np.arange(u.size)
    in_pos = np.isin(j, ui)
    g_u = np.searchsorted(uu, u)
    g_i = uu.size + np.searchsorted(ui, i)
    g_j = np.where(in_pos, uu.size + np.searchsorted(ui, j), uu.size + ui.size + np.searchsorted(un, j))
    group = np.concatenate([g_u, g_i, g_j])
    entry = np.concatenate([4 * slot, 4 * slot + 1, 4 * slot + 2])
    opd_a = np.concatenate([i + item_row0, u + user_row0, u + user_row0])
    order = np.lexsort((entry, group))
    n_groups = uu.size + ui.size + un.size
    out = {k: np.zeros(3 * pad, dtype=np.int32) for k in ("seg_end", "seg", "seg_a")}
    out["seg_rows"] = np.full(3 * pad, -1, dtype=np.int32)
    out["seg_rows"][:n_groups] = np.concatenate([uu + user_row0, ui + item_row0, un + item_row0])
    out["seg_end"][:n_groups] = np.cumsum(np.bincount(group, minlength=n_groups))
    out["seg"][:entry.size] = entry[order]
    out["seg_a"][:entry.size] = opd_a[order]
    out["seg_b"] = np.zeros(pad, dtype=np.int32)
    users_first = order[:u.size]                       # the user groups' entries are the first `cnt` of the batch
    out["seg_b"][:u.size] = (j + item_row0)[entry[users_first] >> 2]
    out["n_uniq_u"], out["n_uniq_i"], out["n_uniq_n"] = (np.array([n], dtype=np.int32) for n in (uu.size, ui.size, un.size))
    return out
