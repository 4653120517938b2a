from ldpc_amd.monte_carlo_simulation.mcs import MonteCarloBscSimulation

__all__ = ["MonteCarloBscSimulation"]
