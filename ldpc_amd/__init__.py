"""ldpc_amd -- MI355X-native batched belief propagation behind the ``ldpc`` decoder API.

The names the reference's package root exports (src_python/ldpc/__init__.py:5-15) that exist here; the decoders outside
the batched-BP path (BpLsdDecoder, BeliefFindDecoder, UnionFindDecoder) are not part of this library.  Resolved lazily:
importing the package does not load the HIP library.
"""
_EXPORTS = {
    "BpDecoder": ("ldpc_amd.bp_decoder", "BpDecoder"),
    "SoftInfoBpDecoder": ("ldpc_amd.bp_decoder", "SoftInfoBpDecoder"),
    "BpOsdDecoder": ("ldpc_amd.bposd_decoder", "BpOsdDecoder"),
    "SinterBpOsdDecoder": ("ldpc_amd.sinter_decoders", "SinterBpOsdDecoder"),
}
# The reference also rebinds the names `bp_decoder` / `bposd_decoder` at the root to the ldpc v1 classes, shadowing the
# sub-packages of the same name; here they stay sub-packages and the v1 classes are `ldpc_amd.bp_decoder.bp_decoder` and
# `ldpc_amd.bposd_decoder.bposd_decoder` (also importable from `ldpc_amd._legacy_ldpc_v1`).


def __getattr__(name):
    if name in _EXPORTS:
        import importlib
        module, attr = _EXPORTS[name]
        return getattr(importlib.import_module(module), attr)
    raise AttributeError(f"module 'ldpc_amd' has no attribute {name!r}")


__all__ = sorted(_EXPORTS)


# ---- opt-in name-level drop-in: `import ldpc` served by this package ---------------------------------------------------
_ALIAS_VERSION = "2.4.1+ldpc_amd"  # the reference release whose API this package mirrors


class _LdpcAliasFinder:
    """Meta-path finder that answers ``import ldpc.<sub>`` with the module ``ldpc_amd.<sub>`` itself (one module object
    under two names: classes, isinstance checks and module state are shared, nothing is imported twice)."""

    def find_spec(self, fullname, path=None, target=None):
        if fullname != "ldpc" and not fullname.startswith("ldpc."):
            return None
        import importlib.machinery
        real = "ldpc_amd" + fullname[len("ldpc"):]
        try:
            import importlib.util
            if fullname != "ldpc" and importlib.util.find_spec(real) is None:
                return None
        except (ImportError, ValueError):
            return None
        return importlib.machinery.ModuleSpec(fullname, self, origin=f"alias of {real}", is_package=True)

    def create_module(self, spec):
        import importlib
        if spec.name == "ldpc":
            return _make_alias_root()
        module = importlib.import_module("ldpc_amd" + spec.name[len("ldpc"):])
        self._own_spec = getattr(self, "_own_spec", {})
        self._own_spec[id(module)] = module.__spec__  # (the import machinery is about to overwrite it with the alias spec)
        return module

    def exec_module(self, module):  # the module is already initialised under its own name; give it its own spec back
        own = getattr(self, "_own_spec", {}).pop(id(module), None)
        if own is not None:
            module.__spec__ = own


def _make_alias_root():
    """What the reference's package root binds (src_python/ldpc/__init__.py:1-15), as far as it exists here: the decoder
    classes, ``__version__``, and the ldpc v1 classes under the names ``bp_decoder`` / ``bposd_decoder`` -- which, as in the
    reference, shadow the sub-packages of the same name as ATTRIBUTES of the root (``import ldpc.bp_decoder`` and
    ``from ldpc.bp_decoder import BpDecoder`` go through ``sys.modules`` and keep working)."""
    import importlib
    import sys
    import types
    root = types.ModuleType("ldpc", "alias of ldpc_amd (ldpc_amd.install_as_ldpc): " + (__doc__ or "").splitlines()[0])
    root.__path__ = []  # a package; its sub-modules come from _LdpcAliasFinder
    root.__version__ = _ALIAS_VERSION
    sys.modules.setdefault("ldpc", root)
    for sub in ("bp_decoder", "bposd_decoder", "sinter_decoders"):
        sys.modules["ldpc." + sub] = importlib.import_module("ldpc_amd." + sub)
    for name, (module, attr) in _EXPORTS.items():
        setattr(root, name, getattr(importlib.import_module(module), attr))
    root.bp_decoder = importlib.import_module("ldpc_amd.bp_decoder").bp_decoder        # "Legacy syntax", __init__.py:13-15
    root.bposd_decoder = importlib.import_module("ldpc_amd.bposd_decoder").bposd_decoder
    return root


def install_as_ldpc(force: bool = False):
    """Make ``import ldpc`` (and ``ldpc.bp_decoder``, ``ldpc.bposd_decoder``, ``ldpc.codes``, ``ldpc.noise_models``,
    ``ldpc.monte_carlo_simulation``, ``ldpc.sinter_decoders``, ``ldpc.ckt_noise``, ``ldpc.helpers`` ...) resolve to this
    package, so that code written against the reference runs unchanged::

        import ldpc_amd; ldpc_amd.install_as_ldpc()
        from ldpc import BpDecoder, BpOsdDecoder          # the MI355X decoders
        from ldpc.codes import rep_code

    Opt-in and refused when a real ``ldpc`` distribution is importable (unless ``force=True``): shadowing an installed
    package silently would be a trap.  Returns the alias root module.  Decoders this library does not build (BpLsdDecoder,
    BeliefFindDecoder, UnionFindDecoder; reference ``__init__.py:7,9,11``) stay absent: ``from ldpc import BpLsdDecoder``
    is an ImportError, not a silent substitute.
    """
    import importlib
    import importlib.util
    import sys
    for f in sys.meta_path:
        if isinstance(f, _LdpcAliasFinder):
            return importlib.import_module("ldpc")
    already = sys.modules.get("ldpc")
    if already is None:
        try:
            spec = importlib.util.find_spec("ldpc")
        except (ImportError, ValueError):
            spec = None
        real_present = spec is not None
    else:
        real_present = True
    if real_present and not force:
        raise RuntimeError("a real `ldpc` package is importable; ldpc_amd.install_as_ldpc(force=True) would shadow it")
    if force:
        for name in [k for k in sys.modules if k == "ldpc" or k.startswith("ldpc.")]:
            del sys.modules[name]
    sys.meta_path.insert(0, _LdpcAliasFinder())
    return importlib.import_module("ldpc")


__all__ = sorted([*_EXPORTS, "install_as_ldpc"])
