"""``ldpc_amd.bposd_decoder`` -- drop-in for ``ldpc.bposd_decoder`` (BP + OSD-0) on MI355X."""
from ldpc_amd.bposd_decoder._bposd_decoder import BpOsdDecoder

__all__ = ["BpOsdDecoder"]
