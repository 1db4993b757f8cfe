"""Make the reference's own model files run on this engine without editing them.

SELFRec model files import ``base.graph_recommender``, ``base.torch_interface``, ``util.sampler``,
``util.loss_torch``, ``data.augmentor`` ... by those top-level names.  ``install()`` registers this
package's mirrors under exactly those names in ``sys.modules``; the reference's ``model/`` directory
stays on ``sys.path`` and is imported as is:

    import selfrec_amd.dropin as dropin
    dropin.install()
    sys.path.insert(0, "/path/to/SELFRec")          # for model/graph/XSimGCL.py only
    from model.graph.XSimGCL import XSimGCL           # unmodified reference file, HIP kernels underneath
"""
import importlib
import sys

MIRRORED = {
    "base": ["recommender", "graph_recommender", "torch_interface"],
    "data": ["loader", "data", "graph", "ui_graph", "augmentor"],
    "util": ["conf", "sampler", "loss_torch", "algorithm", "evaluation", "logger"],
}


def install(overwrite: bool = True) -> None:
    for pkg, mods in MIRRORED.items():
        mirror = importlib.import_module(f"{__package__}.{pkg}")
        if overwrite or pkg not in sys.modules:
            sys.modules[pkg] = mirror
        for m in mods:
            sys.modules[f"{pkg}.{m}"] = importlib.import_module(f"{__package__}.{pkg}.{m}")
    # model files call .cuda() on modules / tensors and use numba only through util.algorithm,
    # which the mirror does not need: nothing else to patch


def uninstall() -> None:
    for pkg, mods in MIRRORED.items():
        for m in mods:
            sys.modules.pop(f"{pkg}.{m}", None)
        if getattr(sys.modules.get(pkg), "__name__", "").startswith(__package__):
            sys.modules.pop(pkg, None)
