"""Seam B1 ON THE MI355X: the class body pyprob_amd/binding.py puts under the real pyprob -
`pyprob_amd.hip_network._HipNetworkMixin`: parameter re-binding into the flat HBM buffer, `_polymorph` growth, `_loss` as an
autograd.Function over `pyprob_hip::ic_loss`, `HipAdam` inside the torch optimizer protocol, pickling, `_infer_init` /
`_infer_step` - driven with TRAINING SESSIONS RECORDED FROM THE STOCK REFERENCE (tests/golden/make_session.py ran pyprob's own
`Model.learn_inference_network` / `posterior` on its own network and wrote down every minibatch, loss, initial and final
weight). The reference is Python and may not travel to the GPU box in any form; its module tree is rebuilt here from the
recorded names and values (tests/binding_standin.py: data holders, no arithmetic). What must come out on the device: the stock
loss trajectory, the stock final weights and optimizer moments, `grad is None` where the reference's autograd leaves it, the
proposal log-probabilities of `_infer_step` (through the float64 oracle, which is itself checked against the stock network's
recorded proposals on the same weights)."""
import pytest

from session_checks import check_grad_none_set, check_infer_steps, check_pickle_roundtrip, check_training_session, load_session

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')


@pytest.mark.parametrize('case', ['gum', 'gumm', 'ffcat', 'gumm2'])
def test_recorded_training_session_through_the_mixin_on_the_device(case):
    net, meta, arrays = check_training_session(case, 'cuda:0')
    assert net._hip_engine.params.is_cuda and all(p.is_cuda for p in net.parameters())
    clone = check_pickle_roundtrip(net, meta, arrays)
    final = load_session(case)[3]
    assert check_infer_steps(clone, meta, arrays, final) >= 24


def test_grad_none_for_the_addresses_a_minibatch_does_not_visit_on_the_device():
    visited, known = check_grad_none_set('gumm', 'cuda:0')
    assert visited < known


def test_one_rank_distributed_sync_of_the_mixin():
    """`_distributed_sync_grad` / `_distributed_sync_parameters` / `_distributed_update_train_loss` of the mixin
    (inference_network.py:281-333 as one all-reduce of [gradients | presence]) on a one-rank RCCL group: the gradients and the
    presence map survive the exchange, the optimizer divides by the world size."""
    import os
    import torch.distributed as dist
    from binding_standin import new_network, session_batch
    net, meta, arrays, init, final = new_network('gumm', 'cuda:0')
    for i in range(meta['iterations']):
        net._polymorph(session_batch(meta, arrays, i))
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29533')
    created = not dist.is_initialized()
    if created:
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda:0'))
    try:
        net._optimizer_type, net._learning_rate_init, net._weight_decay, net._momentum = 'ADAM', 1e-3, 0.0, 0.9
        net._create_optimizer()
        net._distributed_sync_parameters()
        batch = session_batch(meta, arrays, 0)
        net._optimizer.zero_grad()
        ok, loss = net._loss(batch)
        loss.backward()
        had = {n: (p.grad is not None) for n, p in net.named_parameters()}
        g0 = net._hip_engine.grads.clone()
        net._distributed_sync_grad(1)
        assert torch.equal(net._hip_engine.grads, g0) and net._hip_grad_scale == 1.0
        assert {n: (p.grad is not None) for n, p in net.named_parameters()} == had
        out = net._distributed_update_train_loss(float(loss), 1)
        assert abs(float(out) - float(loss)) < 1e-6
        net._optimizer.step()
    finally:
        if created:
            dist.destroy_process_group()
