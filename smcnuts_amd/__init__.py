"""smcnuts_amd: the SMC-NUTS hot path on MI355X (gfx950).

Host side mirrors the reference's operator interface
(UoL-SignalProcessingGroup/SMC-NUTS: SMCSampler, Samples, NUTSProposal,
ForwardLKernel, GaussianApproxLKernel, ESSTempering, Estimate, StanModel-shaped
targets); the compute is hand-written HIP behind the C ABI of
include/smcnuts_hip.h (libsmcnuts_hip.so), reached through ctypes.  No PyTorch
on this path and no CPU fallback.
"""
from .smc_sampler import SMCSampler  # noqa: F401
from .model.targets import ArmaModel, GaussianTarget, HostTarget, IsoGaussian, PRMwCDModel, StanModel  # noqa: F401
