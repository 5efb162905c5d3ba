"""The checks of the binding's session replay, shared by the CPU run (oracle-backed operators, tests/test_binding_session.py)
and the MI355X run (tests/test_gpu_binding_session.py): `_HipNetworkMixin` - the class body pyprob_amd/binding.py puts under
the real pyprob - driven with minibatches RECORDED from the stock reference, against what the stock reference computed."""
import io
import math
import warnings

import numpy as np
import torch

from binding_standin import load_session, new_network, replay_training, session_batch
from oracle import ic_oracle as O


def check_training_session(case, device, engine_factory=None, loss_rtol=2e-4, weight_rtol=5e-3, weight_atol=5e-4):
    """pyprob's learn_inference_network on its own network, replayed through the mixin: same parameter set in the same
    order at every growth step, the stock loss trajectory, the stock final weights, the stock per-address iteration
    counters, `grad is None` exactly where the reference's autograd leaves it, Adam's first moment of two tensors."""
    net, meta, arrays, init, final = new_network(case, device, engine_factory)
    losses = replay_training(net, meta, arrays)
    names = [n for n, _ in net.named_parameters()]
    assert names == meta['param_order']                                   # registration order of the reference's module tree
    assert net._history_num_params == meta['history_num_params']
    np.testing.assert_allclose(losses, meta['losses'], rtol=loss_rtol, atol=loss_rtol)
    eng = net._hip_engine
    # Final weights, per element. Adam normalises the gradient: an element whose gradient is at fp32 round-off level (a dead
    # ReLU unit's weights, ...) may move by +-lr per step in either run (the sign of m / sqrt(v) is noise there) - no element
    # may differ by more than that both ways, and all but a few per cent of a tensor / one per cent of the network must agree
    # within the trajectory tolerance (measured on MI355X against the fp32 reference: 3.3 % of one head's first layer)
    step_bound = 2.0 * meta['learning_rate'] * meta['iterations'] * 1.01
    loose_total, n_total = 0, 0
    for name, p in net.named_parameters():
        assert isinstance(p, torch.nn.Parameter) and p.data_ptr() == eng.tensor(name).data_ptr(), name      # a view of the flat buffer
        assert str(p.device) == str(eng.device)
        got, want = p.detach().cpu().numpy().astype(np.float64), final[name].astype(np.float64)
        diff = np.abs(got - want)
        assert diff.max() <= step_bound, (name, diff.max())
        loose = diff > weight_atol + weight_rtol * np.abs(want)
        assert loose.mean() <= 0.08, (name, loose.mean(), diff.max())
        loose_total += int(loose.sum())
        n_total += loose.size
    assert loose_total <= 0.01 * n_total, (loose_total, n_total)
    for a, layer in net._layers_proposal.items():
        assert layer._total_train_iterations == meta['total_train_iterations'][a], a
    # the optimizer state in torch.optim.Adam's per-parameter format (what pyprob's _save pickles, inference_network.py:170-186)
    sd = net._optimizer.state_dict()
    by_param = {id(p): k for k, p in enumerate(net._optimizer.param_groups[0]['params'])}
    params = dict(net.named_parameters())
    for k in meta['exp_avg_watch']:
        st = sd['state'][by_param[id(params[names[k]])]]
        np.testing.assert_allclose(st['exp_avg'].cpu().numpy().reshape(-1), arrays['exp_avg_%d' % k].reshape(-1), rtol=2e-3, atol=2e-6)
        np.testing.assert_allclose(st['exp_avg_sq'].cpu().numpy().reshape(-1), arrays['exp_avg_sq_%d' % k].reshape(-1), rtol=4e-3, atol=1e-9)
    return net, meta, arrays


def check_grad_none_set(case, device, engine_factory=None):
    """One recorded minibatch of the LAST growth step: parameters of addresses the minibatch does not visit keep
    `grad is None` (the presence map of `_distributed_sync_grad`, inference_network.py:300-315)."""
    net, meta, arrays, init, final = new_network(case, device, engine_factory)
    for i in range(meta['iterations']):
        net._polymorph(session_batch(meta, arrays, i))
    lens = [int(arrays['b%d_trace_len' % i].max()) for i in range(meta['iterations'])]
    i = int(np.argmin(lens))
    batch = session_batch(meta, arrays, i)
    visited = {v.address for tr in batch.traces for v in tr.variables_controlled}
    ok, loss = net._loss(batch)
    assert ok and loss.requires_grad and loss.dim() == 0
    loss.backward()
    for name, p in net.named_parameters():
        owner = [a for a in meta['addresses'] if ('.' + a + '.') in name + '.' and not name.startswith('_layers_address_embedding')]
        if owner and name.startswith('_layers_proposal.') and owner[0] not in visited:
            assert p.grad is None, name
        if name.startswith('_layers_observe_embedding') or name == '_layers_lstm.weight_ih_l0':
            assert p.grad is not None and float(p.grad.abs().max()) > 0, name
    return len(visited), len(meta['addresses'])


def check_pickle_roundtrip(net, meta, arrays):
    """torch.save / torch.load of the module (pyprob's _save / _load pickle the network object): the engine is dropped,
    rebuilt from the module tree on first use, and computes the same loss."""
    batch = session_batch(meta, arrays, 0)
    with torch.no_grad():
        _, before = net._loss(batch)
    buf = io.BytesIO()
    torch.save(net, buf)
    buf.seek(0)
    clone = torch.load(buf, weights_only=False)
    assert clone._hip_engine is None and type(clone).__name__ == type(net).__name__
    with torch.no_grad():
        _, after = clone._loss(batch)
    assert clone._hip_engine is not None and clone._hip_engine is not net._hip_engine
    assert abs(float(after) - float(before)) <= 1e-6 * abs(float(before))
    for (n0, p0), (n1, p1) in zip(net.named_parameters(), clone.named_parameters()):
        assert n0 == n1 and torch.equal(p0.detach().cpu(), p1.detach().cpu())
    return clone


def check_infer_steps(net, meta, arrays, final, logq_rtol=1e-4):
    """`_infer_init` + `_infer_step` of the mixin the way `state.sample` drives them (pyprob/state.py:203-219), one particle at
    a time, on the program's control flow: every proposal log-prob the network returns equals the float64 oracle's for the
    value the network drew; and the oracle itself reproduces the STOCK network's recorded proposal log-probs on the particles
    the reference drew with these final weights (so the chain device -> oracle -> stock reference is closed on this network)."""
    onet = O.Net({n: np.asarray(v, np.float64) for n, v in final.items()}, meta['obs_names'], K=meta['mixture_components'])
    obs = np.array([float(meta['observe'][n]) for n in meta['obs_names']])
    _, logq_ref, _, _ = (O.is_rescore if meta.get('network', 'lstm') == 'lstm' else O.is_rescore_feedforward)(onet, obs, arrays['is_trace_len'], arrays['is_addr_idx'], arrays['is_values'], arrays['is_prior'],
                                     meta['addresses'], meta['dist_names'])
    np.testing.assert_allclose(np.asarray(logq_ref).reshape(-1), arrays['is_logq'], rtol=2e-4, atol=2e-4)
    # the device weights of `net` are the replayed ones (within the trajectory tolerance of the stock's): score against THEM
    dnet = O.Net({n: p.detach().cpu().numpy().astype(np.float64) for n, p in net.named_parameters()}, meta['obs_names'],
                 K=meta['mixture_components'])
    from binding_standin import _distribution
    from pyprob_amd.trace import Variable
    torch.manual_seed(4)
    net._infer_init({n: torch.tensor(float(v)) for n, v in meta['observe'].items()})
    trace_len, addr_idx, values, prior, logq = [], [], [], [], []
    case = meta['case']

    def statement(address, pr, prev):
        """state.sample's IC branch for one variable (state.py:203-219): proposal, draw, log q."""
        a = meta['addresses'].index(address)
        var = Variable(distribution=_distribution(meta['dist_names'][a], pr), address=address, address_base=address, control=True)
        with warnings.catch_warnings():
            warnings.simplefilter('error')                                 # "Using prior" would be a failure here
            proposal = net._infer_step(var, prev_variable=prev)
        value = proposal.sample()
        lq = proposal.log_prob(value, sum=True)
        var.value = torch.as_tensor(value, dtype=torch.float32).reshape(())
        addr_idx.append(a), values.append(float(var.value)), prior.append(list(pr) + [0.0] * (3 - len(pr))), logq.append(float(lq))
        return var

    for particle in range(24):
        prev, n_vars = None, 0
        if case == 'ffcat':                                                # c ~ Categorical; mu ~ Normal(2 c - 1, 1.5)
            c = statement(meta['addresses'][0], (0.2, 0.3, 0.5), None)
            statement(meta['addresses'][1], (float(c.value) * 2.0 - 1.0, 1.5), c)
            n_vars = 2
        elif case == 'gum':
            statement(meta['addresses'][0], (1.0, math.sqrt(5.0)), None)
            n_vars = 1
        else:                                                              # the Marsaglia loop of the program
            while 2 * (n_vars // 2) + 2 <= len(meta['addresses']):         # (not deeper than any address training saw)
                x = statement(meta['addresses'][n_vars], (-1.0, 1.0), prev)
                prev = statement(meta['addresses'][n_vars + 1], (-1.0, 1.0), x)
                n_vars += 2
                if float(x.value) ** 2 + float(prev.value) ** 2 < 1.0:
                    break
        trace_len.append(n_vars)
    rescore = O.is_rescore if meta.get('network', 'lstm') == 'lstm' else O.is_rescore_feedforward
    _, want, _, _ = rescore(dnet, obs, np.asarray(trace_len), np.asarray(addr_idx), np.asarray(values, np.float64),
                                 np.asarray(prior, np.float64), meta['addresses'], meta['dist_names'])
    np.testing.assert_allclose(logq, np.asarray(want).reshape(-1), rtol=logq_rtol, atol=logq_rtol)
    return len(logq)


__all__ = ['check_training_session', 'check_grad_none_set', 'check_pickle_roundtrip', 'check_infer_steps', 'load_session']
