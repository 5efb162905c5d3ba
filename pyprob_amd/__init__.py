"""pyprob_amd: MI355X-native inference-compilation engine behind pyprob's InferenceNetworkLSTM / importance-sampling
API. Hand-written HIP (gfx950) behind a C ABI (include/pyprob_amd.h); PyTorch-ROCm is used for HBM allocation,
streams and torch.distributed only."""
__version__ = '0.1.0'
