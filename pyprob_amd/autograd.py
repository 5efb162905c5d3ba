"""The autograd / optimizer face of the flat-buffer engine: what lets a `torch.nn.Module` whose parameters are views of
an `ICEngine`'s flat HBM buffer be trained by ordinary PyTorch code (`loss.backward()`, `optimizer.step()`).

A "network" here is any object with
    _hip_engine               the ICEngine (flat params / grads / Adam moments, network description, operator handle)
    _hip_named_parameters()   [(state_dict name, nn.Parameter)] - every parameter's `.data` is `engine.tensor(name)`
    _hip_grads_clean          bool, the flat gradient buffer is all zero
    _hip_grad_scale           float, multiplied into the gradients by the next optimizer step (1 / world size)
pyprob_amd/binding.py provides it on top of pyprob's InferenceNetwork classes. No pyprob import here: the GPU tests drive
these classes with a plain nn.Module (tests/test_gpu_binding.py)."""
import torch

from . import lib as L
from .ops import ops


def presence(network, bring_home=False):
    """Per engine tensor (the order of the flat buffer, not of `parameters()`): does the parameter have a gradient
    (`grad is not None`, the reference's presence map inference_network.py:300). bring_home: a gradient that does not
    alias the flat buffer (autograd cloned it, or the user replaced it) is copied into its slot."""
    eng = network._hip_engine
    index = {n: i for i, n in enumerate(eng.spec.tensors.keys())}
    present = [False] * len(index)
    for name, p in network._hip_named_parameters():
        if p.grad is None:
            continue
        present[index[name]] = True
        if bring_home:
            flat = eng.tensor(name, eng.grads)
            if p.grad.data_ptr() != flat.data_ptr():
                flat.copy_(p.grad)
    return tuple(present)


class HipLoss(torch.autograd.Function):
    """loss = `pyprob_hip::ic_loss`(flat parameters, packed minibatch); d loss / d parameter = views of the flat gradient
    buffer the same call filled. Inputs: the participating parameters (names in `names`)."""

    @staticmethod
    def forward(ctx, network, packed, names, *params):
        eng = network._hip_engine
        flags = L.PP_LOSS_BACKWARD | (0 if network._hip_grads_clean else L.PP_LOSS_ZERO_GRADS)
        batch_dev, batch_host = packed.op_tensors(eng.device)
        eng._ensure_workspace(packed.n_traces, packed.n_rows)
        loss, status, _ = ops.ic_loss(eng.params, eng.grads, eng.workspace, batch_dev, batch_host, eng.net_handle, flags)
        network._hip_grads_clean = False
        network._hip_status = status
        ctx.network, ctx.names = network, names
        return loss.reshape(())

    @staticmethod
    def backward(ctx, grad_output):
        eng = ctx.network._hip_engine
        # d(c * loss) / d parameter = c * (what the kernel wrote): one pass over the flat buffer (c = 1 for loss.backward();
        # multiplying by exactly 1.0 changes no bit, and reading c to skip the pass would synchronise the stream)
        eng.grads.mul_(grad_output.reshape(()).to(eng.grads.device))
        return (None, None, None) + tuple(eng.tensor(n, eng.grads) for n in ctx.names)


class _HipOptimizer(torch.optim.Optimizer):
    """Common part of the flat-buffer optimizers: the presence map (`grad is None` -> the tensor is not stepped, like
    torch's per-parameter loop), the LARC wrapper (pyprob/nn/optimizer_larc.py, constructed with its defaults at
    inference_network.py:351-352) as a gradient rewrite before the step, zero_grad by `grad = None`."""

    def __init__(self, network, defaults, larc=False):
        self._network = network
        self._larc = bool(larc)
        super().__init__(list(network.parameters()), defaults)
        self._active_cache = {}
        self._stepped = set()       # engine tensor indices that have optimizer state

    def _begin(self):
        net = self._network
        eng = net._hip_engine
        present = presence(net, bring_home=True)
        act = self._active_cache.get(present)
        if act is None:
            act = self._active_cache[present] = torch.tensor(present, dtype=torch.float32).to(eng.device)
            self._stepped.update(i for i, f in enumerate(present) if f)
        eng.active.copy_(act)
        eng._active_key = None          # (ICEngine.loss caches which presence map is in place)
        group = self.param_groups[0]
        wd, scale = float(group['weight_decay']), float(net._hip_grad_scale)
        if self._larc:
            need = L.larc_scratch_floats(eng.params.numel(), eng.active.numel())
            if getattr(self, '_larc_scratch', None) is None or self._larc_scratch.numel() < need:
                self._larc_scratch = torch.empty(need, dtype=torch.float32, device=eng.params.device)
            ops.larc_scale(eng.params, eng.grads, eng.chunk_tensor, eng.active, float(group['lr']), wd, scale, 0.002, 1e-8,
                           1.0 / 16000.0, True, self._larc_scratch, None)
            wd, scale = 0.0, 1.0          # both are in the gradients now (optimizer_larc.py:81,101)
        return net, eng, group, wd, scale

    def _end(self, net):
        net._hip_grad_scale = 1.0
        net._hip_grads_clean = True       # the consumed gradient chunks were cleared (zero_grad of the next step)

    def zero_grad(self, set_to_none=True):
        # the HIP path tracks "did this parameter take part" by `grad is None`, like the reference's presence map
        for p in self.param_groups[0]['params']:
            p.grad = None

    def _groups_state(self):
        groups = [dict((k, v) for k, v in self.param_groups[0].items() if k != 'params')]
        groups[0]['params'] = list(range(len(self.param_groups[0]['params'])))
        return groups

    def _load_groups(self, state_dict):
        for k, v in state_dict['param_groups'][0].items():
            if k != 'params':
                self.param_groups[0][k] = v


class HipSGD(_HipOptimizer):
    """optim.SGD(lr, momentum, nesterov=True, weight_decay) (inference_network.py:350) as ONE kernel over the network's flat
    buffers; the momentum buffers live in the engine's `exp_avg` buffer. larc=True: Optimizer.SGD_LARC."""

    def __init__(self, network, lr, momentum=0.9, weight_decay=0.0, nesterov=True, larc=False):
        super().__init__(network, dict(lr=lr, momentum=momentum, dampening=0, weight_decay=weight_decay, nesterov=nesterov),
                         larc=larc)

    @torch.no_grad()
    def step(self, closure=None):
        net, eng, group, wd, scale = self._begin()
        ops.sgd_step(eng.params, eng.grads, eng.exp_avg, eng.chunk_tensor, eng.active, float(group['lr']),
                     float(group['momentum']), bool(group['nesterov']), wd, scale, L.PP_ADAM_ZERO_GRADS, None)
        self._end(net)

    def state_dict(self):
        net = self._network
        eng = net._hip_engine
        names = list(eng.spec.tensors.keys())
        state = {}
        if float(self.param_groups[0]['momentum']) != 0.0:
            for i, (name, p) in enumerate(net._hip_named_parameters()):
                if names.index(name) in self._stepped:
                    state[i] = dict(momentum_buffer=eng.tensor(name, eng.exp_avg).detach().cpu().clone())
        return dict(state=state, param_groups=self._groups_state())

    def load_state_dict(self, state_dict):
        net = self._network
        eng = net._hip_engine
        names = list(eng.spec.tensors.keys())
        eng.reset_optimizer()
        self._stepped = set()
        for i, (name, p) in enumerate(net._hip_named_parameters()):
            st = state_dict['state'].get(i)
            if st is None or st.get('momentum_buffer') is None:
                continue
            eng.tensor(name, eng.exp_avg).copy_(st['momentum_buffer'].reshape(p.shape))
            self._stepped.add(names.index(name))
        self._load_groups(state_dict)


class HipAdam(_HipOptimizer):
    """optim.Adam(lr, weight_decay) (inference_network.py:348) as ONE kernel over the network's flat buffers.
    larc=True: Optimizer.ADAM_LARC."""

    def __init__(self, network, lr, weight_decay=0.0, betas=(0.9, 0.999), eps=1e-8, larc=False):
        super().__init__(network, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay), larc=larc)

    @torch.no_grad()
    def step(self, closure=None):
        net, eng, group, wd, scale = self._begin()
        b1, b2 = group['betas']
        ops.adam_step(eng.params, eng.grads, eng.exp_avg, eng.exp_avg_sq, eng.chunk_tensor, eng.active, eng.tensor_step,
                      eng.arrived, float(group['lr']), float(b1), float(b2), float(group['eps']), wd, scale,
                      L.PP_ADAM_ZERO_GRADS, None)
        self._end(net)

    def state_dict(self):
        net = self._network
        eng = net._hip_engine
        steps = eng.tensor_step.cpu().tolist()
        names = list(eng.spec.tensors.keys())
        state = {}
        for i, (name, p) in enumerate(net._hip_named_parameters()):
            k = names.index(name)
            if steps[k] > 0:
                state[i] = dict(step=torch.tensor(float(steps[k])), exp_avg=eng.tensor(name, eng.exp_avg).detach().cpu().clone(),
                                exp_avg_sq=eng.tensor(name, eng.exp_avg_sq).detach().cpu().clone())
        return dict(state=state, param_groups=self._groups_state())

    def load_state_dict(self, state_dict):
        net = self._network
        eng = net._hip_engine
        names = list(eng.spec.tensors.keys())
        eng.reset_optimizer()
        steps = torch.zeros(len(names), dtype=torch.int32)
        for i, (name, p) in enumerate(net._hip_named_parameters()):
            st = state_dict['state'].get(i)
            if st is None:
                continue
            eng.tensor(name, eng.exp_avg).copy_(st['exp_avg'].reshape(p.shape))
            eng.tensor(name, eng.exp_avg_sq).copy_(st['exp_avg_sq'].reshape(p.shape))
            steps[names.index(name)] = int(st['step'])
        eng.tensor_step.copy_(steps)
        eng.moments_written()
        self._load_groups(state_dict)
