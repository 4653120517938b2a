"""ldpc v1 call syntax on top of the v2 mirrors (reference: src_python/ldpc/_legacy_ldpc_v1/)."""
from ldpc_amd._legacy_ldpc_v1._legacy import bp_decoder, bposd_decoder

__all__ = ["bp_decoder", "bposd_decoder"]
