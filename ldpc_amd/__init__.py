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
