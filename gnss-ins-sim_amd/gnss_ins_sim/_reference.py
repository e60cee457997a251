"""Fall-through to a checkout of the reference for everything that is OFF the accelerated path.

This package shadows the reference's ``gnss_ins_sim`` and ``demo_algorithms`` packages (same names, ahead on ``sys.path``),
but provides only the hot path of SURVEY.md section 8.  A user plugin that calls, say, ``attitude.quat_update``
(attitude.py:665) or ``attitude.dcm2quat`` (:294) -- as the reference's own demo_algorithms/inclinometer_mahony.py:115,151
does -- must not break because of the shadowing.  When ``$GNSS_INS_SIM_REFERENCE`` names a checkout of the reference:

* a module this package does not have (``gnss_ins_sim.kml_gen``, ``gnss_ins_sim.psd.time_series_from_psd``,
  ``gnss_ins_sim.geoparams.geomag``, ``demo_algorithms.inclinometer_mahony`` ...) is imported from the checkout under its usual
  name (a meta-path finder that only answers for names under these two packages which this package lacks);
* a NAME that one of this package's modules does not define (``attitude.quat_update``) is looked up in the reference module of the
  same name (module ``__getattr__`` -> ``delegate``), loaded once under a private name inside the same package so that its
  relative imports (``from ..attitude import attitude``) resolve to this package's modules -- hot-path functions therefore stay
  the accelerated ones even when reference code calls them.

Nothing is picked up from ``sys.path`` or the current directory: the checkout is named explicitly by the environment variable.
No bytecode is written into the checkout (it may be read-only and is not ours to write into).  Without the variable the
package behaves as before: the missing name raises AttributeError / ImportError, and the message says how to get it.
"""
import importlib.abc
import importlib.machinery
import importlib.util
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
_PKG_ROOT = os.path.dirname(_HERE)                       # holds gnss_ins_sim/ and demo_algorithms/
_TOPS = ('gnss_ins_sim', 'demo_algorithms')
ENV = 'GNSS_INS_SIM_REFERENCE'


def reference_root():
    """The checkout named by $GNSS_INS_SIM_REFERENCE, or None (unset, missing, or this very tree)."""
    root = os.environ.get(ENV)
    if not root:
        return None
    root = os.path.abspath(root)
    if not os.path.isdir(os.path.join(root, 'gnss_ins_sim')) or os.path.samefile(root, _PKG_ROOT):
        return None
    return root


class _NoBytecodeLoader(importlib.machinery.SourceFileLoader):
    """SourceFileLoader that never writes a .pyc next to the source."""

    def set_data(self, path, data, *, _mode=0o666):      # noqa: D401 -- the cache write of SourceLoader.get_code
        return None


def _spec(fullname, path, package_dir=None):
    loader = _NoBytecodeLoader(fullname, path)
    return importlib.util.spec_from_file_location(fullname, path, loader=loader,
                                                  submodule_search_locations=[package_dir] if package_dir else None)


class ReferenceFinder(importlib.abc.MetaPathFinder):
    """Answers only for ``gnss_ins_sim.*`` / ``demo_algorithms.*`` names that this package does NOT provide."""

    def find_spec(self, fullname, path=None, target=None):
        parts = fullname.split('.')
        if parts[0] not in _TOPS or len(parts) < 2:
            return None
        ours = os.path.join(_PKG_ROOT, *parts)
        if os.path.isfile(ours + '.py') or os.path.isfile(os.path.join(ours, '__init__.py')):
            return None                                  # this package has it: the normal machinery loads ours
        root = reference_root()
        if root is None:
            return None
        cand = os.path.join(root, *parts)
        if os.path.isfile(os.path.join(cand, '__init__.py')):
            return _spec(fullname, os.path.join(cand, '__init__.py'), cand)
        if os.path.isfile(cand + '.py'):
            return _spec(fullname, cand + '.py')
        return None


def install():
    """Put the finder on sys.meta_path once (called by the two package __init__ files)."""
    if not any(isinstance(f, ReferenceFinder) for f in sys.meta_path):
        sys.meta_path.insert(0, ReferenceFinder())


_DELEGATES = {}


def delegate(module_name, attr):
    """``attr`` of the REFERENCE module that ``module_name`` (e.g. 'gnss_ins_sim.attitude.attitude') shadows.  Raises
    AttributeError when there is no checkout or the reference does not have the name either."""
    if attr.startswith('__'):
        raise AttributeError(attr)
    ref = _DELEGATES.get(module_name)
    if ref is None:
        root = reference_root()
        path = None if root is None else os.path.join(root, *module_name.split('.')) + '.py'
        if path is None or not os.path.isfile(path):
            raise AttributeError("module %r of the MI355X drop-in provides the accelerated path only and has no attribute %r; "
                                 "name a checkout of the reference in $%s to fall through to its implementation"
                                 % (module_name, attr, ENV))
        parent, _, leaf = module_name.rpartition('.')
        private = '%s._reference_%s' % (parent, leaf)    # inside the same package: its relative imports resolve to OUR modules
        spec = _spec(private, path)
        ref = importlib.util.module_from_spec(spec)
        sys.modules[private] = ref
        try:
            spec.loader.exec_module(ref)
        except BaseException:
            sys.modules.pop(private, None)
            raise
        _DELEGATES[module_name] = ref
    try:
        return getattr(ref, attr)
    except AttributeError:
        raise AttributeError('neither the drop-in nor the reference module %r has attribute %r' % (module_name, attr)) from None
