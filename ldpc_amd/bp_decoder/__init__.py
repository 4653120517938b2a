"""``ldpc_amd.bp_decoder`` -- drop-in for ``ldpc.bp_decoder`` on MI355X.

Exports the names the reference module exports (bp_decoder/__init__.py:1-7): ``BpDecoder``, ``SoftInfoBpDecoder``
(serial soft-syndrome min-sum, bp.hpp:547-660), ``BpDecoderBase``, ``io_test`` and the ldpc v1 syntax ``bp_decoder``.
"""
from ldpc_amd.bp_decoder._bp_decoder import BpDecoder, BpDecoderBase, SoftInfoBpDecoder, io_test


def __getattr__(name):  # `bp_decoder` subclasses BpDecoder and lives in a package that imports this one: resolve lazily
    if name == "bp_decoder":
        from ldpc_amd._legacy_ldpc_v1._legacy import bp_decoder
        return bp_decoder
    raise AttributeError(name)


__all__ = ["BpDecoder", "SoftInfoBpDecoder", "BpDecoderBase", "io_test", "bp_decoder"]
