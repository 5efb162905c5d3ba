"""pyprob_amd: MI355X-native inference-compilation engine behind pyprob's InferenceNetworkLSTM / importance-sampling
API. Hand-written HIP (gfx950) behind a C ABI (include/pyprob_amd.h); PyTorch-ROCm is used for HBM allocation,
streams and torch.distributed only."""
__version__ = '0.1.0'


def __getattr__(name):
    """Lazy pyprob-style top level: pyprob_amd.sample / observe / Model / InferenceEngine / PriorInflation ... (importing the package
    must not require torch or a GPU)."""
    if name in ('sample', 'observe', 'TraceMode', 'InferenceEngine', 'PriorInflation', 'InferenceNetwork', 'LearningRateScheduler',
                'Optimizer'):
        from . import state
        value = getattr(state, name)
        globals()[name] = value          # (resolved once: `pyprob.sample(...)` in a program runs per statement and per path)
        return value
    if name == 'Model':
        from .model import Model
        globals()[name] = Model
        return Model
    raise AttributeError(name)
