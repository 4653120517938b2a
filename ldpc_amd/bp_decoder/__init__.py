"""``ldpc_amd.bp_decoder`` -- drop-in for ``ldpc.bp_decoder`` on MI355X.

Exports the names the reference module exports (bp_decoder/__init__.py:1-7): ``BpDecoder``, ``SoftInfoBpDecoder``
(serial soft-syndrome min-sum, bp.hpp:547-660), ``BpDecoderBase``, ``io_test``.
"""
from ldpc_amd.bp_decoder._bp_decoder import BpDecoder, BpDecoderBase, SoftInfoBpDecoder, io_test

__all__ = ["BpDecoder", "SoftInfoBpDecoder", "BpDecoderBase", "io_test"]
