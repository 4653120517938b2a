"""``bp_decoder`` / ``bposd_decoder``: the ldpc v1 constructor syntax the reference keeps for old scripts
(_legacy_ldpc_v1/_legacy_bp_decoder.py:6-139, _legacy_bposd_decoder.py:6-160; exported from ``ldpc.bp_decoder`` /
``ldpc.bposd_decoder`` and from the package root).  In the reference they are Python subclasses of the Cython classes
whose ``__init__`` re-validates the v1 keywords after the base ``__cinit__`` has consumed the same keywords; here the
base initialiser is called explicitly with them.  Decoding itself is the parent class's (device path)."""
from __future__ import annotations

import warnings

import numpy as np

from ldpc_amd.bp_decoder._bp_decoder import _MS_ALIASES, _PS_ALIASES, BpDecoder, BpDecoderBase
from ldpc_amd.bposd_decoder._bposd_decoder import BpOsdDecoder


def _v1_channel(parity_check_matrix, error_rate, channel_probs):
    """v1 rule: ``channel_probs`` (if its first entry is not None) wins over ``error_rate``; its length must be n."""
    n = parity_check_matrix.shape[1]
    if channel_probs[0] is not None:
        if len(channel_probs) != n:
            raise ValueError(f"The length of the channel probability vector must be eqaul to the block length n={n}.")
        return np.array([channel_probs[j] for j in range(n)], dtype=float)
    if error_rate != 0:
        return None
    raise ValueError("Either the error_rate or channel_probs must be specified.")


def _v1_bp_method(bp_method) -> str:
    key = str(bp_method).lower()
    if key in _PS_ALIASES:
        return "ps"
    if key in _MS_ALIASES:
        return "ms"
    raise ValueError(f"BP method '{bp_method}' is invalid.\
                            Please choose from the following methods:'product_sum',\
                            'minimum_sum'")


class bp_decoder(BpDecoder):
    """Legacy ldpc_v1 syntax for :class:`BpDecoder` (``parity_check_matrix``, ``error_rate``, ``max_iter``, ``bp_method``,
    ``ms_scaling_factor``, ``channel_probs``, ``input_vector_type``)."""

    def __init__(self, parity_check_matrix, error_rate=None, max_iter=0, bp_method="ps", ms_scaling_factor=1.0,
                 channel_probs=[None], input_vector_type="auto", error_channel=None, **kwargs):
        BpDecoderBase.__init__(self, parity_check_matrix, error_rate=error_rate, max_iter=max_iter, bp_method=bp_method,
                               ms_scaling_factor=ms_scaling_factor, channel_probs=channel_probs, error_channel=error_channel,
                               **kwargs)
        warnings.warn("This is the old syntax for the `bp_decoder` from `ldpc v1`. Use the `BpDecoder` class from `ldpc v2` for additional features.")
        error_channel = _v1_channel(parity_check_matrix, error_rate, channel_probs)
        if type(input_vector_type) is int and input_vector_type == -1:
            input_vector_type = "auto"
        elif not (type(input_vector_type) is str and input_vector_type in ("auto", "syndrome", "received_vector")):
            raise Exception(f"TypeError: input_vector type must be either 'syndrome', 'received_vector' or 'auto'. Not {input_vector_type}")
        self.bp_method = _v1_bp_method(bp_method)
        self.max_iter = int(max_iter)
        self.error_channel = error_channel
        self.error_rate = error_rate
        self.ms_scaling_factor = float(ms_scaling_factor)
        self.input_vector_type = input_vector_type

    @property
    def channel_probs(self):
        return self.error_channel

    def update_channel_probs(self, channel):
        self.error_channel = channel


class bposd_decoder(BpOsdDecoder):
    """Legacy ldpc_v1 syntax for :class:`BpOsdDecoder` (adds ``osd_method`` -- 'osd_0' | 'osd_e' | 'osd_cs' and the v1
    numeric aliases -- and ``osd_order``)."""

    def __init__(self, parity_check_matrix, error_rate=None, max_iter=0, bp_method="ps", ms_scaling_factor=1.0,
                 channel_probs=[None], osd_method="osd_0", osd_order=0, **kwargs):
        warnings.warn("This is the old syntax for the `bposd_decoder` from `ldpc v1`. Use the `BpOsdDecoder` class from `ldpc v2` for additional features.")
        bp_method = _v1_bp_method(bp_method)
        key = str(osd_method).lower()
        if key in ["osd_0", "0", "osd0"]:
            osd_method, osd_order = "osd_0", 0
        elif key in ["osd_e", "1", "osde", "exhaustive", "e"]:
            osd_method = "osd_e"
            if osd_order > 15:
                print("WARNING: Running the 'OSD_E' (Exhaustive method) with search depth greater than 15 is not recommended. Use the 'osd_cs' method instead.")
        elif key in ["osd_cs", "2", "osdcs", "combination_sweep", "cs"]:
            osd_method = "osd_cs"
        else:
            raise ValueError(f"ERROR: OSD method '{osd_method}' invalid. Please choose from the following methods: 'OSD_0', 'OSD_E' or 'OSD_CS'.")
        error_channel = _v1_channel(parity_check_matrix, error_rate, channel_probs)
        init = dict(max_iter=int(max_iter), bp_method=bp_method, ms_scaling_factor=float(ms_scaling_factor), osd_method=osd_method,
                    osd_order=osd_order)
        if error_channel is not None:
            init["error_channel"] = list(error_channel)
        else:
            init["error_rate"] = error_rate
        BpOsdDecoder.__init__(self, parity_check_matrix, **init, **kwargs)

    @property
    def channel_probs(self):
        return self.error_channel

    def update_channel_probs(self, channel):
        self.error_channel = channel
