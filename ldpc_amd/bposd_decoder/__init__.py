"""``ldpc_amd.bposd_decoder`` -- drop-in for ``ldpc.bposd_decoder`` (BP + OSD-0 / OSD-E / OSD-CS) on MI355X.

``BpOsdDecoder`` and the ldpc v1 syntax ``bposd_decoder`` (reference bposd_decoder/__init__.py:1-2; its
``SoftInfoBpOsdDecoder`` is not provided)."""
from ldpc_amd.bposd_decoder._bposd_decoder import BpOsdDecoder


def __getattr__(name):
    if name == "bposd_decoder":
        from ldpc_amd._legacy_ldpc_v1._legacy import bposd_decoder
        return bposd_decoder
    raise AttributeError(name)


__all__ = ["BpOsdDecoder", "bposd_decoder"]
