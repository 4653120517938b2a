"""sinter adaptors of the overlapping-window decoders (reference: ckt_noise/sinter_overlapping_window_decoder.py:14-130).

``sinter`` only needs objects with ``compile_decoder_for_dem`` / ``decode_via_files`` and ``decode_shots_bit_packed``;
the classes derive from ``sinter.Decoder`` / ``sinter.CompiledDecoder`` when sinter is installed and are plain classes
with the same methods otherwise.  Shot files are read and written with this package's own b8 reader (no stim).
"""
from __future__ import annotations

import pathlib

import numpy as np

from ldpc_amd.ckt_noise.bposd_overlapping_window import BpOsdOverlappingWindowDecoder
from ldpc_amd.ckt_noise.dem_text import load_dem
from ldpc_amd.sinter_decoders.sinter_bposd_decoder import read_b8, write_b8

try:  # pragma: no cover - sinter is not part of this image
    from sinter import CompiledDecoder as _CompiledBase, Decoder as _DecoderBase
except ImportError:
    _CompiledBase = _DecoderBase = object


class SinterCompiledDecoder_OWD_Base(_CompiledBase):
    """Wraps a window decoder that implements ``decode_batch`` (reference :14-33)."""

    def __init__(self, decoder):
        self.decoder = decoder

    def decode_shots_bit_packed(self, *, bit_packed_detection_event_data: np.ndarray) -> np.ndarray:
        return self.decoder.decode_batch(shots=bit_packed_detection_event_data, bit_packed_shots=True, bit_packed_predictions=True)


class SinterDecoder_Base_OWD(_DecoderBase):
    def __init__(self, Decoder_cls, **decoder_kwargs):
        self.Decoder_cls = Decoder_cls
        self.decoder_kwargs = decoder_kwargs

    def compile_decoder_for_dem(self, *, dem):
        return SinterCompiledDecoder_OWD_Base(self.Decoder_cls(dem, **self.decoder_kwargs))

    def decode_via_files(self, *, num_shots: int, num_dets: int, num_obs: int, dem_path: pathlib.Path,
                         dets_b8_in_path: pathlib.Path, obs_predictions_b8_out_path: pathlib.Path, tmp_dir: pathlib.Path) -> None:
        """Reference :52-105: the model at ``dem_path`` configures the decoder, shots in / predictions out in b8."""
        dem = load_dem(pathlib.Path(dem_path))
        decoder = self.Decoder_cls(dem, **self.decoder_kwargs)
        packed = read_b8(dets_b8_in_path, dem.num_detectors, num_shots)
        predictions = decoder.decode_batch(packed, bit_packed_shots=True, bit_packed_predictions=True)
        nbytes = (dem.num_observables + 7) // 8
        write_b8(obs_predictions_b8_out_path, np.ascontiguousarray(predictions[:, :nbytes]))


class SinterDecoder_BPOSD_OWD(SinterDecoder_Base_OWD):
    def __init__(self, **decoder_kwargs):
        super().__init__(BpOsdOverlappingWindowDecoder, **decoder_kwargs)
