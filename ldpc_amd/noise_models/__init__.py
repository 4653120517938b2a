from ldpc_amd.noise_models.bsc import generate_bsc_error, generate_bsc_batch

__all__ = ["generate_bsc_error", "generate_bsc_batch"]
