"""In-memory loader for the *reference* hot-path modules (build container only).

TEST INFRASTRUCTURE -- never imported by the product path.  Only
``oracle/make_golden.py`` and the container-side timing script use it.

The reference (``/root/reference``) is Python 2.7 source.  It is read as text,
shimmed *in memory* (nothing is written into this repository) and executed into
fresh module objects registered under the reference's own module names so that
its intra-package imports (``import probability_functions as prob`` ...) resolve.

Shims:
  1. ``print`` statement -> function (lib2to3 ``fix_print``);
  2. Py2 integer division ``/`` -> ``//`` at vp_localisation.py:133,157,158;
  3. ``np.array(to_be_removed)`` -> ``dtype=int`` at vp_localisation.py:329,394
     (NumPy 2 rejects an empty float index in ``np.delete``);
  4. ``affinity='precomputed'`` -> ``metric='precomputed'`` at vp_localisation.py:575
     (scikit-learn >= 1.4);
  5. sphere_mapping.py: ``Image.fromstring/tostring`` -> ``frombytes/tobytes`` (:11),
     ``np.fromstring`` -> ``np.frombuffer(...).copy()`` (:28),
     ``set_axis_bgcolor`` -> ``set_facecolor`` (:49,:92).

``/root/reference`` does not exist on the GPU box; importing this module there
raises immediately.
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("VPK_REFERENCE_ROOT", "/root/reference")

_PRINT_FIX = None


def _fix_print(src, name):
    global _PRINT_FIX
    from lib2to3 import refactor
    if _PRINT_FIX is None:
        _PRINT_FIX = refactor.RefactoringTool(["lib2to3.fixes.fix_print"])
    return str(_PRINT_FIX.refactor_string(src if src.endswith("\n") else src + "\n", name))


def _replace(src, old, new, count=None, name=""):
    n = src.count(old)
    if n == 0 or (count is not None and n != count):
        raise RuntimeError("shim for %s: expected %s occurrence(s) of %r, found %d"
                           % (name, count if count is not None else ">=1", old, n))
    return src.replace(old, new)


def _shim_vp_localisation(src):
    src = _replace(src, "sphere[ra*sA/rA:(ra+1)*sA/rA, rb*sB/rB:(rb+1)*sB/rB]",
                   "sphere[ra*sA//rA:(ra+1)*sA//rA, rb*sB//rB:(rb+1)*sB//rB]", 1, "vp:133")
    src = _replace(src, "max_response[0] + ra*sA/rA", "max_response[0] + ra*sA//rA", 1, "vp:157")
    src = _replace(src, "max_response[1] + rb*sB/rB", "max_response[1] + rb*sB//rB", 1, "vp:158")
    src = _replace(src, "to_be_removed = np.array(to_be_removed)",
                   "to_be_removed = np.array(to_be_removed, dtype=int)", 2, "vp:329,394")
    src = _replace(src, "affinity='precomputed'", "metric='precomputed'", 1, "vp:575")
    return src


def _shim_sphere_mapping(src):
    src = _replace(src, 'Image.fromstring("RGBA", (w, h), buf.tostring())',
                   'Image.frombytes("RGBA", (w, h), buf.tobytes())', 1, "sm:11")
    src = _replace(src, "np.fromstring(fig.canvas.tostring_argb(), dtype=np.uint8)",
                   "np.frombuffer(fig.canvas.tostring_argb(), dtype=np.uint8).copy()", 1, "sm:28")
    src = _replace(src, "set_axis_bgcolor", "set_facecolor", 2, "sm:49,92")
    return src


_SHIMS = {"vp_localisation": _shim_vp_localisation, "sphere_mapping": _shim_sphere_mapping}

# load order matters: dependencies first
_MODULES = ["coordinate_conversion", "probability_functions", "vp_localisation",
            "calc_horizon", "auc", "sphere_mapping"]


def load_reference(modules=None, quiet=True):
    """Return {name: module} for the shimmed reference hot-path modules."""
    if not os.path.isdir(REFERENCE_ROOT):
        raise RuntimeError("reference tree %s is absent (expected on the GPU box): the "
                           "reference oracle only runs in the build container" % REFERENCE_ROOT)
    import numpy  # noqa: F401
    import numpy.matlib  # noqa: F401  (probability_functions.py:74 uses np.matlib unimported)
    out = {}
    for name in (modules or _MODULES):
        if name in sys.modules and getattr(sys.modules[name], "__vpk_ref__", False):
            out[name] = sys.modules[name]
            continue
        path = os.path.join(REFERENCE_ROOT, name + ".py")
        with open(path) as fh:
            src = fh.read()
        src = _fix_print(src, name)
        if name in _SHIMS:
            src = _SHIMS[name](src)
        mod = types.ModuleType(name)
        mod.__file__ = path
        mod.__vpk_ref__ = True
        if quiet:
            mod.__dict__["print"] = lambda *a, **k: None
        sys.modules[name] = mod
        exec(compile(src, path, "exec"), mod.__dict__)
        out[name] = mod
    return out


if __name__ == "__main__":
    mods = load_reference()
    print("loaded:", ", ".join(sorted(mods)))
