"""echoseal_amd -- MI355X (gfx950) implementation of the EchoSeal receive hot path.

Host mirror of the reference's `rtwm` interface (detector, polar_fast, fastpolar, embedder, utils,
crypto) over hand-written HIP kernels (echoseal_amd/csrc, C ABI in include/echoseal_hip.h).
"""
__version__ = "0.1.0"
