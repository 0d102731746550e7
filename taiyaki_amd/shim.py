"""Make code written against the reference's module names run on the HIP operators.

    import taiyaki_amd.shim; taiyaki_amd.shim.install()
    from taiyaki import ctc, layers, decode            # now the gfx950 operators

`install()` handles both situations a Taiyaki user can be in:

* the reference package `taiyaki` is importable (its Python layers, file readers and CLI are
  wanted as they are): its hot-path entry points are re-pointed --
  `taiyaki.ctc` (the Cython extension module, taiyaki/ctc/__init__.py:1) is replaced by
  `taiyaki_amd.ctc`, `taiyaki.layers.flipflop_logpartition` / `log_partition_flipflop` (layers.py:1875-1890, 1277-1299),
  `taiyaki.decode.flipflop_viterbi` / `flipflop_make_trans` (decode.py:15-72),
  `taiyaki.qscores.errprobs_from_trans` (qscores.py:88-142),
  `taiyaki.flipflop_remap.flipflop_remap` (flipflop_remap.py:6-88) and
  `taiyaki.decodeutil.beamsearch` / `forward` / `backward` (decodeutil/decodeutil.pyx:9-108) by their
  HIP counterparts;
* it is not (this repository on its own): a package `taiyaki` is registered whose submodules
  ARE the taiyaki_amd ones, so `bin/train_flipflop.py`-shaped callers resolve every name of
  the hot path (`ctc`, `layers`, `decode`, `flipflopfings`, `flipflop_remap`, `qscores`,
  `decodeutil`, `basecall_helpers`, `maths.RollingMAD`).

`uninstall()` restores what was there.  Nothing here computes: it is name plumbing, and the
operators it installs still refuse CPU tensors (no fallback).
"""
import importlib
import importlib.util
import sys
import types

_SAVED = {}
_PATCHED = []

_SUBMODULES = {
    "ctc": "taiyaki_amd.ctc",
    "layers": "taiyaki_amd.layers",
    "decode": "taiyaki_amd.decode",
    "flipflopfings": "taiyaki_amd.flipflopfings",
    "flipflop_remap": "taiyaki_amd.flipflop_remap",
    "qscores": "taiyaki_amd.qscores",
    "decodeutil": "taiyaki_amd.decodeutil",
    "basecall_helpers": "taiyaki_amd.basecall_helpers",
}
_FUNCTIONS = [
    ("layers", "flipflop_logpartition", "taiyaki_amd.layers"),
    ("layers", "log_partition_flipflop", "taiyaki_amd.layers"),
    ("layers", "global_norm_flipflop", "taiyaki_amd.layers"),
    ("decode", "flipflop_viterbi", "taiyaki_amd.decode"),
    ("decode", "flipflop_make_trans", "taiyaki_amd.decode"),
    ("qscores", "errprobs_from_trans", "taiyaki_amd.qscores"),
    ("flipflop_remap", "flipflop_remap", "taiyaki_amd.flipflop_remap"),
    ("decodeutil", "beamsearch", "taiyaki_amd.decodeutil"),
    ("decodeutil", "forward", "taiyaki_amd.decodeutil"),
    ("decodeutil", "backward", "taiyaki_amd.decodeutil"),
]


def _reference_importable():
    if "taiyaki" in sys.modules and getattr(sys.modules["taiyaki"], "__taiyaki_amd_shim__", False):
        return False
    try:
        return importlib.util.find_spec("taiyaki") is not None
    except (ImportError, ValueError):
        return False


def install(force_standalone=False):
    """Returns "patched" (reference package re-pointed) or "standalone" (shim package registered)."""
    if _SAVED or _PATCHED:
        uninstall()
    if not force_standalone and _reference_importable():
        pkg = importlib.import_module("taiyaki")
        # the Cython extension: `from taiyaki import ctc` / `import taiyaki.ctc`
        amd_ctc = importlib.import_module("taiyaki_amd.ctc")
        for key in ("taiyaki.ctc", "taiyaki.ctc.ctc"):
            _SAVED[key] = sys.modules.get(key)
            sys.modules[key] = amd_ctc
        _PATCHED.append((pkg, "ctc", getattr(pkg, "ctc", None)))
        pkg.ctc = amd_ctc
        for sub, name, src in _FUNCTIONS:
            try:
                mod = importlib.import_module("taiyaki." + sub)
            except ImportError:          # optional dependency of that reference module missing
                continue
            _PATCHED.append((mod, name, getattr(mod, name, None)))
            setattr(mod, name, getattr(importlib.import_module(src), name))
        return "patched"
    pkg = types.ModuleType("taiyaki")
    pkg.__doc__ = "taiyaki_amd.shim: the reference's hot-path module names on the gfx950 operators"
    pkg.__path__ = []
    pkg.__taiyaki_amd_shim__ = True
    _SAVED["taiyaki"] = sys.modules.get("taiyaki")
    sys.modules["taiyaki"] = pkg
    for sub, src in _SUBMODULES.items():
        mod = importlib.import_module(src)
        key = "taiyaki." + sub
        _SAVED[key] = sys.modules.get(key)
        sys.modules[key] = mod
        setattr(pkg, sub, mod)
    maths = types.ModuleType("taiyaki.maths")
    clipping = importlib.import_module("taiyaki_amd.clipping")
    maths.RollingMAD, maths.med_mad, maths.MAD_SD_FACTOR = clipping.RollingMAD, clipping.med_mad, clipping.MAD_SD_FACTOR
    _SAVED["taiyaki.maths"] = sys.modules.get("taiyaki.maths")
    sys.modules["taiyaki.maths"] = maths
    pkg.maths = maths
    return "standalone"


def uninstall():
    for obj, name, old in reversed(_PATCHED):
        if old is None:
            try:
                delattr(obj, name)
            except AttributeError:
                pass
        else:
            setattr(obj, name, old)
    _PATCHED.clear()
    for key, old in _SAVED.items():
        if old is None:
            sys.modules.pop(key, None)
        else:
            sys.modules[key] = old
    _SAVED.clear()
