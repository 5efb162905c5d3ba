"""GPU twin of tests/test_binding_reference.py: the operators (`torch.ops.pyprob_hip.*`), the autograd.Function and the
optimizer of the reference-side binding on the device, against the goldens recorded from the reference. pyprob itself is
not on the GPU box; a plain nn.Module whose parameters are views of the engine's flat buffer stands in for the
InferenceNetworkLSTM subclass (the binding's parameter protocol, pyprob_amd/autograd.py)."""
import numpy as np
import pytest

from conftest import load_golden
from helpers import engine_from_golden, grad_check, packed_from_golden, rel_err
from oracle import ic_oracle as O

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')


def bound_module(eng):
    """nn.Module with one nn.Parameter per engine tensor (reference state_dict names), `.data` = the flat view."""
    class Net(torch.nn.Module):
        _hip_grads_clean = False
        _hip_grad_scale = 1.0

        def __init__(self):
            super().__init__()
            self._hip_engine = eng
            self._names = []
            for i, name in enumerate(eng.spec.tensors):
                p = torch.nn.Parameter(torch.empty(0, device=eng.device))
                p.data = eng.tensor(name)
                self.register_parameter('p%d' % i, p)
                self._names.append(name)

        def _hip_named_parameters(self):
            return list(zip(self._names, self.parameters()))
    return Net()


def test_autograd_loss_gives_golden_gradients_and_none_for_untouched_parameters(golden):
    from pyprob_amd.autograd import HipLoss
    case, meta, params, batch, loss, isr = golden
    eng = engine_from_golden(meta, params)
    net = bound_module(eng)
    pb = packed_from_golden(meta, batch, eng.spec)
    act = eng.spec.active_mask(pb.cur_counts, pb.prev_counts)
    named = net._hip_named_parameters()
    taking_part = [(n, p) for (n, p), a in zip(named, act) if a > 0]
    out = HipLoss.apply(net, pb, [n for n, _ in taking_part], *[p for _, p in taking_part])
    assert out.dim() == 0 and out.requires_grad
    assert abs(float(out) - float(loss['loss'])) <= 2e-5 * abs(float(loss['loss']))
    assert int(net._hip_status.item()) == 0
    out.backward()
    has_grad = dict(zip(meta['param_names'], meta['has_grad']))
    gold = {n: loss['g%d' % i] for i, n in enumerate(meta['param_names'])}
    for name, p in named:
        if not has_grad[name]:
            assert p.grad is None, name                     # the reference's autograd leaves it None (presence map)
            continue
        assert p.grad is not None, name
        grad_check('binding_%s/%s' % (case, name), p.grad.cpu().numpy(), gold[name], 1e-5, 5e-8)
    # d(3 loss): the Function scales the flat buffer
    for p in net.parameters():
        p.grad = None
    out = HipLoss.apply(net, pb, [n for n, _ in taking_part], *[p for _, p in taking_part])
    (3.0 * out).backward()
    for name, p in named:
        if has_grad[name]:
            grad_check('binding3_%s/%s' % (case, name), p.grad.cpu().numpy(), 3.0 * gold[name], 1e-5, 1.5e-7)


def test_hip_adam_inside_the_torch_optimizer_protocol():
    """HipAdam.step / zero_grad / state_dict / load_state_dict + a LambdaLR scheduler, against the oracle's Adam on the
    oracle's gradients (weights after three steps), on the GUMM golden (parameters without a gradient keep step 0)."""
    from pyprob_amd.autograd import HipAdam, HipLoss
    meta, params, batch, loss, isr = load_golden('gumm')
    eng = engine_from_golden(meta, params)
    net = bound_module(eng)
    pb = packed_from_golden(meta, batch, eng.spec)
    act = eng.spec.active_mask(pb.cur_counts, pb.prev_counts)
    named = net._hip_named_parameters()
    taking_part = [(n, p) for (n, p), a in zip(named, act) if a > 0]
    opt = HipAdam(net, lr=1e-3, weight_decay=0.0)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lambda it: 1.0 / (1 + it))
    P = {k: v.astype(np.float64).copy() for k, v in params.items()}
    M = {k: np.zeros_like(v) for k, v in P.items()}
    V = {k: np.zeros_like(v) for k, v in P.items()}
    for step in range(1, 4):
        opt.zero_grad()
        out = HipLoss.apply(net, pb, [n for n, _ in taking_part], *[p for _, p in taking_part])
        out.backward()
        opt.step()
        ref = O.loss_and_grads(O.Net(P, meta['obs_names'], K=10), batch, meta['addresses'], meta['dist_names'])
        assert abs(float(out) - ref['loss']) < 1e-4 * abs(ref['loss'])
        for (n, _), a in zip(named, act):
            if a > 0:
                O.adam_step(P[n], ref['grads'][n], M[n], V[n], step, 1e-3 / step)
        sched.step()
    sd = eng.state_dict()
    assert max(rel_err(sd[n].numpy(), P[n]) for n in P) < 5e-3
    steps = eng.tensor_step.cpu().numpy()
    assert set(steps.tolist()) == {0, 3} and (steps > 0).tolist() == (act > 0).tolist()
    state = opt.state_dict()
    assert len(state['state']) == int((act > 0).sum()) and state['param_groups'][0]['lr'] == pytest.approx(1e-3 / 4)
    moments = eng.exp_avg.clone()
    eng.reset_optimizer()
    opt.load_state_dict(state)
    assert torch.equal(eng.exp_avg, moments) and eng.tensor_step.cpu().numpy().tolist() == steps.tolist()


def test_operators_reject_bad_arguments_and_cpu_tensors():
    from pyprob_amd import ops as P
    meta, params, batch, loss, isr = load_golden('gum')
    eng = engine_from_golden(meta, params)
    pb = packed_from_golden(meta, batch, eng.spec)
    bdev, bhost = pb.op_tensors(eng.device)
    eng._ensure_workspace(pb.n_traces, pb.n_rows)
    small = torch.empty(16, dtype=torch.uint8, device=eng.device)
    with pytest.raises(RuntimeError, match='workspace too small'):
        P.ops.ic_loss(eng.params, eng.grads, small, bdev, bhost, eng.net_handle, 0)
    with pytest.raises(RuntimeError, match='float32'):
        P.ops.ic_loss(eng.params.double(), eng.grads, eng.workspace, bdev, bhost, eng.net_handle, 0)
    with pytest.raises(NotImplementedError):      # no CPU kernel in the product (the CPU suite registers test doubles itself)
        P.ops.log_prob(0, torch.zeros(4, device='meta'), 0, torch.ones(4, device='meta'), 0, torch.zeros(4, device='meta'), 4)
    l0, st, lp = P.ops.ic_loss(eng.params, eng.grads, eng.workspace, bdev, bhost, eng.net_handle, 4)   # forward, keep lp
    assert abs(float(l0) - float(loss['loss'])) <= 1e-4 * abs(float(loss['loss'])) and lp.numel() == pb.n_rows
