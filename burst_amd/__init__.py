"""burst_amd -- MI355X-native alignment hot path with the behaviour of knights-lab/BURST.

Only what the path needs lives here: csrc/ (HIP kernels, the C ABI of include/burst_hip.h and the C host),
capi.py (ctypes binding of the C ABI), synth.py (seeded synthetic genomes/reads for tests and the bench).
The device library has no CPU fallback: importing burst_amd.capi without a built libburst_hip.so raises.
"""
__version__ = "0.1.0"
