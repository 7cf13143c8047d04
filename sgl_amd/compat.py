"""`sgl` namespace alias: let code written against the reference (PKU-DAIR/SGL) import the MI355X path under the reference's OWN
module names, so that `sgl.models.homo.{SGC, GAMLP, NAFS, ...}` consume it unchanged -- not one line of their files is edited.

    import sgl_amd.compat
    sgl_amd.compat.install()                                   # the plugin API under its reference names
    from sgl.operators.graph_op import LaplacianGraphOp        # -> sgl_amd.operators.graph_op.LaplacianGraphOp
    from sgl.operators.message_op import LearnableWeightedMessageOp

    sgl_amd.compat.install(reference_root="/path/to/SGL")      # additionally: the reference's model FILES become importable
    from sgl.models.homo.gamlp import GAMLP                     # executes the reference's sgl/models/homo/gamlp.py as it is;
                                                                # its four imports (gamlp.py:1-4) resolve to sgl_amd modules

What is registered in sys.modules (the modules a reference model file imports, sgl/models/homo/*.py:1-4, plus their parents):

    sgl.operators, sgl.operators.base_op, sgl.operators.utils, sgl.operators.graph_op, sgl.operators.message_op   -> sgl_amd.operators.*
    sgl.models.base_model, sgl.models.simple_models                                                                 -> sgl_amd.models.*
    sgl, sgl.models, sgl.models.homo       namespace shells (created only if the name is not imported yet); with reference_root their
                                           __path__ points INTO the reference checkout, so `sgl.models.homo.<file>` is found there --
                                           without running the reference's package __init__ files (sgl/__init__.py pulls in the
                                           dataset / tasks packages and their third-party dependencies; sgl/models/homo/__init__.py
                                           is broken at HEAD: sgc_dist.py:1 imports a name from the wrong module)

Nothing of the reference is copied or shipped: with reference_root the files are executed from where they lie.  `uninstall()`
removes exactly what install() added."""
import importlib
import os
import sys
import types

_ALIASES = {
    "sgl.operators": "sgl_amd.operators",
    "sgl.operators.base_op": "sgl_amd.operators.base_op",
    "sgl.operators.utils": "sgl_amd.operators.utils",
    "sgl.operators.graph_op": "sgl_amd.operators.graph_op",
    "sgl.operators.message_op": "sgl_amd.operators.message_op",
    "sgl.models.base_model": "sgl_amd.models.base_model",
    "sgl.models.simple_models": "sgl_amd.models.simple_models",
}
_SHELLS = {"sgl": "sgl", "sgl.models": os.path.join("sgl", "models"), "sgl.models.homo": os.path.join("sgl", "models", "homo")}
_installed = []           # names this module put into sys.modules


def install(reference_root=None, force=False):
    """Register the aliases.  reference_root: a checkout of the reference (the directory that holds its `sgl/` package) whose model
    files should be importable as sgl.models.homo.<name>.  A real `sgl` package that is ALREADY imported is left alone unless
    force=True (then the seven aliased submodules replace its own).  Returns the list of module names that now resolve to sgl_amd."""
    if reference_root is not None:
        reference_root = os.path.abspath(reference_root)
        if not os.path.isdir(os.path.join(reference_root, "sgl", "models", "homo")):
            raise FileNotFoundError(f"{reference_root!r} does not hold sgl/models/homo")
    real = sys.modules.get("sgl")
    if real is not None and not getattr(real, "__sgl_amd_shell__", False) and not force:
        raise RuntimeError("a real `sgl` package is already imported; call install() before importing it, or install(force=True) to "
                           "replace its operator / base-model modules with the sgl_amd ones")
    for name, rel in _SHELLS.items():
        mod = sys.modules.get(name)
        if mod is None:
            mod = types.ModuleType(name)
            mod.__sgl_amd_shell__ = True
            mod.__path__ = []
            mod.__package__ = name
            sys.modules[name] = mod
            _installed.append(name)
            parent, _, leaf = name.rpartition(".")
            if parent:
                setattr(sys.modules[parent], leaf, mod)
        if reference_root is not None and getattr(mod, "__sgl_amd_shell__", False):
            path = os.path.join(reference_root, rel)
            if path not in mod.__path__:
                mod.__path__.append(path)
    done = []
    for alias, target in _ALIASES.items():
        mod = importlib.import_module(target)
        if sys.modules.get(alias) is not mod:
            sys.modules[alias] = mod
            _installed.append(alias)
        parent, _, leaf = alias.rpartition(".")
        setattr(sys.modules[parent], leaf, mod)
        done.append(alias)
    return done


def uninstall():
    """remove what install() registered (modules imported THROUGH the aliases, e.g. sgl.models.homo.sgc, go too)"""
    for name in list(sys.modules):
        if name.startswith("sgl.models.homo.") and getattr(sys.modules.get("sgl.models.homo"), "__sgl_amd_shell__", False):
            del sys.modules[name]
    for name in reversed(_installed):
        sys.modules.pop(name, None)
    _installed.clear()


def load_reference_model(name, reference_root):
    """the class `name` (e.g. "GAMLP") of the reference's sgl/models/homo/<file>.py, executed unchanged on top of the sgl_amd
    operators.  File names follow the reference: lower-case class name (GAMLPRecursive -> gamlp_recursive)."""
    install(reference_root)
    fname = {"GAMLPRecursive": "gamlp_recursive", "PASCA_V1": "pasca_v1", "PASCA_V2": "pasca_v2", "PASCA_V3": "pasca_v3"}.get(name, name.lower())
    return getattr(importlib.import_module(f"sgl.models.homo.{fname}"), name)
