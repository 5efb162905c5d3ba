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


class HipAdam(torch.optim.Optimizer):
    """optim.Adam(lr, weight_decay) (inference_network.py:348) as ONE kernel over the network's flat buffers."""

    def __init__(self, network, lr, weight_decay=0.0, betas=(0.9, 0.999), eps=1e-8):
        self._network = network
        super().__init__(list(network.parameters()), dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._active_cache = {}

    @torch.no_grad()
    def step(self, closure=None):
        net = self._network
        eng = net._hip_engine
        group = self.param_groups[0]
        present = presence(net, bring_home=True)
        act = self._active_cache.get(present)
        if act is None:
            act = self._active_cache[present] = torch.tensor(present, dtype=torch.float32).to(eng.device)
        eng.active.copy_(act)
        b1, b2 = group['betas']
        ops.adam_step(eng.params, eng.grads, eng.exp_avg, eng.exp_avg_sq, eng.chunk_tensor, eng.active, eng.tensor_step,
                      eng.arrived, float(group['lr']), float(b1), float(b2), float(group['eps']),
                      float(group['weight_decay']), float(net._hip_grad_scale), L.PP_ADAM_ZERO_GRADS, None)
        net._hip_grad_scale = 1.0
        net._hip_grads_clean = True       # the consumed gradient chunks were cleared (zero_grad of the next step)

    def zero_grad(self, set_to_none=True):
        # the HIP path tracks "did this parameter take part" by `grad is None`, like the reference's presence map
        for p in self.param_groups[0]['params']:
            p.grad = None

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
        groups = [dict((k, v) for k, v in self.param_groups[0].items() if k != 'params')]
        groups[0]['params'] = list(range(len(self.param_groups[0]['params'])))
        return dict(state=state, param_groups=groups)

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
        for k, v in state_dict['param_groups'][0].items():
            if k != 'params':
                self.param_groups[0][k] = v
