from ldpc_amd.sinter_decoders.sinter_bposd_decoder import SinterBpOsdDecoder, decode_b8_files, read_b8, write_b8

__all__ = ["SinterBpOsdDecoder", "decode_b8_files", "read_b8", "write_b8"]
