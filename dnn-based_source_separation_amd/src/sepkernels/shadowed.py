"""
Route A of INTEGRATION.md puts this tree IN FRONT of the reference's `src/` on `sys.path`; the reference's packages are flat
namespace packages, so a module that exists in both trees (`modules/conv.py`, `criterion/distance.py`, ...) resolves here and hides
the reference's file -- including the names this tree has no reason to re-implement (`modules.conv.MultiDilatedConv2d` of the
MMDenseLSTM family, `criterion.pit.ProbPIT`, ...), which other reference modules import.  `fall_through` gives such a module a
PEP 562 `__getattr__` that finds the hidden file further down `sys.path`, loads it under a private name and serves the missing
attribute from it, so `from modules.conv import MultiDilatedConv2d` keeps working in a merged tree.  Without the reference on the
path (the GPU box, a stand-alone install) the attribute is simply missing, as it would be anyway.
The hidden copy RE-DEFINES the classes this tree defines too (SISDR, GlobalLayerNorm ...): objects built from names served here are
the reference's own classes, not this tree's -- isinstance checks against this tree's classes do not see them, and such objects take the
generic (composed) paths.  Only names this tree does not define are ever served, so product code never receives such a twin by accident.
"""
import importlib.util
import os
import sys

_LOADED = {}


def _hidden_module(module_name, own_file):
    if module_name in _LOADED:
        return _LOADED[module_name]
    rel = os.path.join(*module_name.split(".")) + ".py"
    own = os.path.realpath(own_file)
    found = None
    # Where the hidden file may live: SEPK_REFERENCE_SRC (the reference's src/, when set: nowhere else), otherwise the entries of sys.path
    # that were put there explicitly -- never '' / '.' / the working directory: a same-named file lying around where the process happens
    # to run (./criterion/pit.py ...) must not be executed.
    root = os.environ.get("SEPK_REFERENCE_SRC")
    cwd = os.path.realpath(os.getcwd())
    entries = [root] if root else [e for e in sys.path if e not in ("", ".") and os.path.realpath(e) != cwd]
    for entry in entries:
        cand = os.path.join(entry, rel)
        if os.path.isfile(cand) and os.path.realpath(cand) != own:
            spec = importlib.util.spec_from_file_location("_shadowed_." + module_name, cand)
            found = importlib.util.module_from_spec(spec)
            _LOADED[module_name] = found                    # before execution: the hidden file may import the shadowing module back
            try:
                spec.loader.exec_module(found)
            except Exception:
                del _LOADED[module_name]
                raise
            break
    _LOADED[module_name] = found
    return found


def fall_through(module_name, own_file):
    """-> a module-level `__getattr__` for the module `module_name` living in `own_file`"""
    def __getattr__(name):
        if name.startswith("__"):
            raise AttributeError(name)
        hidden = _hidden_module(module_name, own_file)
        if hidden is not None and hasattr(hidden, name):
            return getattr(hidden, name)
        raise AttributeError("module {!r} has no attribute {!r}".format(module_name, name))
    return __getattr__
