"""Make the reference's own model files run on this engine without editing them.

SELFRec model files import ``base.graph_recommender``, ``base.torch_interface``, ``util.sampler``,
``util.loss_torch``, ``data.augmentor`` ... by those top-level names.  ``install()`` registers this
package's mirrors under exactly those names in ``sys.modules``; the reference's ``model/`` directory
stays on ``sys.path`` and is imported as is:

    import selfrec_amd.dropin as dropin
    dropin.install()
    sys.path.insert(0, "/path/to/SELFRec")          # for model/graph/XSimGCL.py only
    from model.graph.XSimGCL import XSimGCL           # unmodified reference file, HIP kernels underneath

``install(fuse=True)`` (or ``SRH_DROPIN_FUSE=1``) goes one step further for the five model files whose whole training
step the fused engine implements (MF, LightGCN, XSimGCL, SimGCL, SGL): when such a file is imported and its bytes -- or,
failing that, its syntax tree without comments and docstrings -- are the reference's (digests below: "unchanged" is checked,
not assumed), the class keeps everything it defines --
``__init__`` (config keys, the torch encoder and its xavier-initialised ``embedding_dict``), ``save``, ``predict``,
``cal_cl_loss`` -- but ``train()`` is served by ``engine.FusedTrainer``: same batches (the global ``random`` stream is
consumed and left as the file's own loop would leave it), same arithmetic (the parity tests of the engine are against
this very file's CPU run), the encoder's parameters aliased to the engine's table so ``save()`` / ``predict()`` /
checkpoints see the trained values.  A file whose CODE was edited keeps its own ``train()`` on the op-level tier, and a line
on stderr says so.
"""
import ast
import hashlib
import importlib
import os
import sys

MIRRORED = {
    "base": ["recommender", "graph_recommender", "torch_interface"],
    "data": ["loader", "data", "graph", "ui_graph", "augmentor"],
    "util": ["conf", "sampler", "loss_torch", "algorithm", "evaluation", "logger"],
}


def _canonical(node):
    """A version-stable dump of a syntax tree: node class names and non-empty fields, no positions, no docstrings -- what
    the file DOES, not how it is typed.  (ast.dump is not used: its text changes between python versions as node classes gain
    optional fields.)"""
    if isinstance(node, ast.AST):
        fields = []
        for name in sorted(node._fields):
            value = getattr(node, name, None)
            if value is None or value == []:
                continue
            if name == "body" and isinstance(value, list) and value and isinstance(value[0], ast.Expr) \
                    and isinstance(getattr(value[0], "value", None), ast.Constant) and isinstance(value[0].value.value, str) \
                    and isinstance(node, (ast.Module, ast.ClassDef, ast.FunctionDef, ast.AsyncFunctionDef)):
                value = value[1:] or [ast.Pass()]                    # a docstring is a comment
            if name in ("kind", "type_comment", "type_params", "type_ignores"):
                continue
            fields.append((name, _canonical(value)))
        return (type(node).__name__, tuple(fields))
    if isinstance(node, list):
        return tuple(_canonical(v) for v in node)
    return repr(node)


def syntax_digest(source: str) -> str:
    """SHA-256 of the canonical syntax dump of a python source text: unchanged by comments, blank lines, line breaks inside
    expressions, quote style or docstrings; changed by any change of code."""
    return hashlib.sha256(repr(_canonical(ast.parse(source))).encode()).hexdigest()


# Digests of the reference's model files at the surveyed commit (Coder-Yu/SELFRec @ 2025-07-25) -- facts about the reference,
# not copies of it (tests/golden/make_fusable_digests.py prints them): model/graph/<name>.py -> (SHA-256 of the bytes, SHA-256
# of the canonical syntax dump).  A file is "the reference's" when EITHER matches: an upstream re-format, a licence header or
# a comment does not drop a user to the op-level tier; a changed statement does -- and says so (maybe_fuse).
FUSABLE = {
    "XSimGCL": ("621ed1ea0181e57ac2924c8ad14f82ceefc7e774080613362ad92c3ae3fe3f2e",
                "88a708970565de7183f3f4356bf2187658c9c636c49d7b05d3bbb6420386f56f"),
    "LightGCN": ("18707b19917c481ea0c1b3f9fcd1db0be2cb0ad0fa955b7affc498d89d52c8dd",
                 "38942f59c276693d5f0834169dd8e4314ebdc50c3f4a5ffb5ce3a1bbc86639db"),
    "SimGCL": ("33d92e815b98f2f19d57fb8615d3076efac7014e16fd66d4aa3c02d870ad6f11",
               "facec86c9e9af60cc11467bd6f7a11fcfbb2c8d849fb7d653b56e48dbff5bf00"),
    "SGL": ("ee53e7821791b7196d44bc84833b694ece64f2cc7d3b5f92dd360fab26c7452a",
            "f4a728dadee8df853608c12bbd811a0dea0d2638d830c8379c0030338c362f72"),
    "MF": ("0564ad7cfc66cca9c79d004efbfa6fda0c51370af266110924a25e260bb563b3",
           "1722940934668b7ab464f1e675debfe23cab13c9c08ea173d6ffd106ca497571"),
}
_state = {"fuse": False, "fused": [], "matched": {}}


def fuse_enabled() -> bool:
    return _state["fuse"] or os.environ.get("SRH_DROPIN_FUSE", "0") not in ("", "0")


def maybe_fuse(cls) -> bool:
    """Called by the mirrored ``GraphRecommender.__init_subclass__`` for every model class: route ``cls.train`` to the
    fused engine iff fusing is on, the class is one of the five the engine implements, it is defined in a module named
    ``model.graph.<its own name>`` and that module's file is the reference's: byte for byte, or -- comments, blank lines,
    formatting and docstrings aside -- statement for statement."""
    name = cls.__name__
    if not fuse_enabled() or name not in FUSABLE or cls.__module__ != f"model.graph.{name}":
        return False
    path = getattr(sys.modules.get(cls.__module__), "__file__", None)
    if not path or not os.path.isfile(path):
        return False
    with open(path, "rb") as f:
        raw = f.read()
    how = "bytes" if hashlib.sha256(raw).hexdigest() == FUSABLE[name][0] else None
    if how is None:
        try:
            how = "syntax" if syntax_digest(raw.decode()) == FUSABLE[name][1] else None
        except (SyntaxError, UnicodeDecodeError):
            how = None
    if how is None:
        # an edited file keeps its own train() on the op-level tier -- said once, not silently
        print(f"[selfrec_amd.dropin] {path}: the code of {name} differs from the reference's -- its own train() runs on the "
              f"op-level tier (HIP kernels under torch autograd), not on the fused engine", file=sys.stderr)
        return False
    _state["matched"][name] = how
    from .model.graph._fused import fused_train_of_reference_class
    cls._reference_train = cls.train
    cls.train = fused_train_of_reference_class
    _state["fused"].append(name)
    return True


def install(overwrite: bool = True, fuse: bool | None = None, fast: bool = True) -> None:
    """``fast`` (default on; ``SRH_DROPIN_FAST=0`` turns it off): the host-side fast paths of the op-level tier --
    ``table[list]`` row gathers as one ``index_select`` on ids uploaded once per batch, ``torch.unique`` of a batch stream
    answered by the sampler, ``torch.optim.Adam`` stepping through the fused kernel (util/fastpath.py).  They apply to ANY
    model file written the reference's way -- an edited copy, a new model -- not only to the five byte-identical ones
    ``fuse`` recognises."""
    if fuse is not None:
        _state["fuse"] = bool(fuse)
    if fast and os.environ.get("SRH_DROPIN_FAST", "1") != "0":
        from .util import fastpath
        fastpath.install()
    for pkg, mods in MIRRORED.items():
        mirror = importlib.import_module(f"{__package__}.{pkg}")
        if overwrite or pkg not in sys.modules:
            sys.modules[pkg] = mirror
        for m in mods:
            sys.modules[f"{pkg}.{m}"] = importlib.import_module(f"{__package__}.{pkg}.{m}")
    # model files call .cuda() on modules / tensors and use numba only through util.algorithm,
    # which the mirror does not need: nothing else to patch


def uninstall() -> None:
    from .util import fastpath
    fastpath.uninstall()
    for pkg, mods in MIRRORED.items():
        for m in mods:
            sys.modules.pop(f"{pkg}.{m}", None)
        if getattr(sys.modules.get(pkg), "__name__", "").startswith(__package__):
            sys.modules.pop(pkg, None)
