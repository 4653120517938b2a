"""``ldpc_amd.bp_decoder`` -- drop-in for ``ldpc.bp_decoder`` on MI355X.

Exports the names the reference module exports (bp_decoder/__init__.py:1-7) for the path this
library replaces: ``BpDecoder``, ``BpDecoderBase``, ``io_test``.  ``SoftInfoBpDecoder`` (serial
soft-syndrome min-sum, bp.hpp:547-665) is outside the hot path and is not provided.
"""
from ldpc_amd.bp_decoder._bp_decoder import BpDecoder, BpDecoderBase, io_test

__all__ = ["BpDecoder", "BpDecoderBase", "io_test"]
