"""Overlay of the native `models` / `utils` packages on a checkout of the reference (joewybean/MMI-Det).

The reference's callers (train.py:24-38, test.py:12-18, detect_twostream.py:11-16) import the hot path AND two dozen
modules that are none of this package's business (utils.datasets, utils.plots, utils.metrics, utils.google_utils,
utils.wandb_logging, models.export ...).  Both live under the top-level names `models` and `utils`, so this package's
`models/` and `utils/` are *overlay packages*:

* whole modules this package does not have (utils.datasets, utils.plots, models.export ...) resolve to the reference's
  files: `extend_package_path` appends <reference>/<pkg> to the package's `__path__`;
* modules both sides have (utils.general, utils.torch_utils, utils.autoanchor, utils.loss, models.common,
  models.experimental, models.yolo_test) resolve to the native ones, and every name the native module does not define
  (utils.general.labels_to_class_weights, utils.torch_utils.select_device, utils.autoanchor.check_anchors, models.common.
  autoShape ...) falls through, lazily (PEP 562 module `__getattr__`), to the reference's module of the same name, which is
  loaded from its file as `<pkg>._reference_<module>`.

The reference checkout is found through $MMIDET_REFERENCE_ROOT, else the first directory on sys.path / the working
directory that holds `models/yolo_test.py` and `utils/datasets.py` and is not this package.  Without one (the GPU box, the
tests) the overlay is inert: unknown names raise AttributeError / ImportError as usual.

`python train.py` puts the script's directory (the reference checkout) at sys.path[0], ahead of PYTHONPATH, so the
interpreter would never see this package's `models/`.  Two ways in, both shown in INTEGRATION.md:

    PYTHONPATH=<repo>/mmi-det_amd python -m mmidet_hip.overlay train.py --cfg ...      # launcher: fixes the order, runs the script
    import mmidet_hip.overlay as o; o.activate()                                        # or: first line of train.py

Nothing here touches the GPU."""
import importlib.util
import os
import sys

PKG_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))      # .../mmi-det_amd
_root_cache = []


def _is_reference_checkout(d):
    return (os.path.isfile(os.path.join(d, 'models', 'yolo_test.py')) and os.path.isfile(os.path.join(d, 'utils', 'datasets.py'))
            and os.path.realpath(d) != os.path.realpath(PKG_ROOT))


def reference_root():
    """Directory of the reference checkout or None."""
    if _root_cache:
        return _root_cache[0]
    env = os.environ.get('MMIDET_REFERENCE_ROOT')
    cands = [env] if env else []
    cands += [p or os.getcwd() for p in sys.path] + [os.getcwd()]
    found = None
    for d in cands:
        if d and os.path.isdir(d) and _is_reference_checkout(d):
            found = os.path.abspath(d)
            break
    if env and found != os.path.abspath(env):
        raise ImportError('MMIDET_REFERENCE_ROOT=%r is not an MMI-Det checkout (models/yolo_test.py, utils/datasets.py)' % env)
    _root_cache.append(found)
    return found


def extend_package_path(pkg_name, pkg_path):
    """Called from models/__init__.py and utils/__init__.py: sub-modules this package lacks come from the reference."""
    root = reference_root()
    if root is not None:
        d = os.path.join(root, pkg_name)
        if os.path.isdir(d) and d not in pkg_path:
            pkg_path.append(d)


def _load_reference_module(pkg_name, mod_name):
    root = reference_root()
    if root is None:
        return None
    alias = '%s._reference_%s' % (pkg_name, mod_name)
    m = sys.modules.get(alias)
    if m is not None:
        return m
    path = os.path.join(root, pkg_name, mod_name + '.py')
    if not os.path.isfile(path):
        return None
    spec = importlib.util.spec_from_file_location(alias, path)
    m = importlib.util.module_from_spec(spec)
    m.__package__ = pkg_name
    sys.modules[alias] = m
    try:
        spec.loader.exec_module(m)
    except BaseException:
        del sys.modules[alias]
        raise
    return m


def fall_through(module_name):
    """-> a module-level __getattr__ for the native module `module_name` (e.g. 'utils.general')."""
    pkg_name, mod_name = module_name.rsplit('.', 1)
    loading = []

    def __getattr__(name):
        if name.startswith('__') or loading:          # (dunder probes by importlib / pickle; re-entrance while loading)
            raise AttributeError(name)
        loading.append(1)
        try:
            ref = _load_reference_module(pkg_name, mod_name)
        finally:
            loading.pop()
        if ref is None or not hasattr(ref, name):
            raise AttributeError('module %r has no attribute %r%s' % (
                module_name, name, '' if ref is not None else ' (and no reference checkout is overlaid: set MMIDET_REFERENCE_ROOT)'))
        return getattr(ref, name)
    return __getattr__


def activate(reference=None):
    """Make `import models...` / `import utils...` resolve to the overlay from now on: this package first on sys.path, the
    reference checkout (argument, $MMIDET_REFERENCE_ROOT or auto-detected) behind it.  Call before the first import of
    either package."""
    for name in ('models', 'utils'):
        m = sys.modules.get(name)
        if m is not None and not os.path.realpath(getattr(m, '__file__', '') or '').startswith(os.path.realpath(PKG_ROOT)):
            raise ImportError('%r was already imported from %s: activate the overlay before the first import' %
                              (name, getattr(m, '__file__', '?')))
    if reference is not None:
        os.environ['MMIDET_REFERENCE_ROOT'] = os.path.abspath(reference)
        del _root_cache[:]
    while PKG_ROOT in sys.path:
        sys.path.remove(PKG_ROOT)
    sys.path.insert(0, PKG_ROOT)
    root = reference_root()
    if root is not None:               # train.py's `import test`, `import global_var`: right behind this package, i.e.
        while root in sys.path:        # ahead of the standard library (which has a `test` package of its own), as the
            sys.path.remove(root)      # script directory is under `python train.py`
        sys.path.insert(1, root)
    return root


def main(argv):
    """python -m mmidet_hip.overlay <script.py> [args...]: run one of the reference's scripts on the overlay."""
    import runpy
    if not argv:
        raise SystemExit('usage: python -m mmidet_hip.overlay <train.py|test.py|detect_twostream.py> [args...]')
    script = os.path.abspath(argv[0])
    here = os.path.dirname(script)
    if 'MMIDET_REFERENCE_ROOT' not in os.environ and _is_reference_checkout(here):
        os.environ['MMIDET_REFERENCE_ROOT'] = here
    activate()
    if here not in sys.path:
        sys.path.insert(1, here)       # what `python script.py` would have put first; now second
    sys.argv = [script] + list(argv[1:])
    runpy.run_path(script, run_name='__main__')


if __name__ == '__main__':
    main(sys.argv[1:])
