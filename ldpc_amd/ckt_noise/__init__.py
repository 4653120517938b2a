"""Circuit-level noise: detector error model -> matrices, overlapping-window decoding (reference: src_python/ldpc/ckt_noise).

Built: ``detector_error_model_to_check_matrices`` (from DEM text, no stim needed), ``BaseOverlappingWindowDecoder``,
``BpOsdOverlappingWindowDecoder``, ``SinterDecoder_BPOSD_OWD``.  Not built: the LSD and PyMatching window decoders (their
inner decoders are outside this package's path), the stim circuit generators (css_code_memory_circuit, edge colouring).
"""
from ldpc_amd.ckt_noise.base_overlapping_window_decoder import BaseOverlappingWindowDecoder, current_round_inds
from ldpc_amd.ckt_noise.dem_matrices import DemMatrices, detector_error_model_to_check_matrices
from ldpc_amd.ckt_noise.bposd_overlapping_window import BpOsdOverlappingWindowDecoder
from ldpc_amd.ckt_noise.sinter_overlapping_window_decoder import SinterDecoder_BPOSD_OWD, SinterDecoder_Base_OWD

__all__ = ["BaseOverlappingWindowDecoder", "current_round_inds", "DemMatrices", "detector_error_model_to_check_matrices",
           "BpOsdOverlappingWindowDecoder", "SinterDecoder_BPOSD_OWD", "SinterDecoder_Base_OWD"]
