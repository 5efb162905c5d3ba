"""CPU ORACLE / BASELINE (test infrastructure, NOT product code): a torch-CPU restatement of one inference-compilation
training step of the reference on the GaussianUnknownMean benchmark network - the modules pyprob builds (SURVEY.md
Appendix B: `nn.Linear` observe embeddings, `nn.LSTM`, `nn.Linear` proposal layers), `loss.backward()` through torch
autograd and `optim.Adam.step()` (pyprob/nn/inference_network.py:348, 486-496), i.e. the kernels the reference really
executes on the host - but vectorised over the minibatch, without the per-trace Python of `Batch` / `_loss`
(pyprob/nn/inference_network_lstm.py:146-196). It is therefore an UPPER bound of the reference's own rate (SURVEY.md §6:
1.76 k traces/s end to end, 8.5 k NN-only on 8 cores) and the honest thing to time on the GPU box, where
/root/reference does not exist. Only bench.py's cpu_baseline leg and tests/ import it.

Pinned: tests/test_oracle.py::test_torch_restatement_matches_the_numpy_oracle (same loss and gradients on the `gum`
golden minibatch, which is itself recorded from the reference)."""
import math

import torch
import torch.nn as nn

FP32_EPS = torch.finfo(torch.float32).eps


class GumNetwork(nn.Module):
    """InferenceNetworkLSTM after _polymorph on a GaussianUnknownMean batch: one address, one distribution type."""

    def __init__(self, lstm_dim=512, K=10, obs_dims=(32, 32), sample_dim=4, address_dim=64, dtype_dim=8):
        super().__init__()
        self.K = K
        self.obs = nn.ModuleList([nn.ModuleList([nn.Linear(1, int((1 + d) / 2)), nn.Linear(int((1 + d) / 2), d)])
                                  for d in obs_dims])                                  # embedding_feedforward.py:24-30
        e = sum(obs_dims)
        self.final = nn.ModuleList([nn.Linear(e, e), nn.Linear(e, e)])                  # inference_network.py:129
        self.lstm = nn.LSTM(e + sample_dim + 2 * (address_dim + dtype_dim), lstm_dim, 1)      # inference_network_lstm.py:30-31
        self.address_embedding = nn.Parameter(torch.zeros(address_dim).normal_())      # :43
        self.dtype_embedding = nn.Parameter(torch.zeros(dtype_dim).normal_())          # :47
        self.sample_embedding = nn.Linear(1, sample_dim)                                 # :54 (unused at the first time step)
        hid = int((lstm_dim + 3 * K) / 2)
        self.proposal = nn.ModuleList([nn.Linear(lstm_dim, hid), nn.Linear(hid, 3 * K)])   # proposal_normal_normal_mixture.py:14
        self.zeros = sample_dim + address_dim + dtype_dim

    def load_reference_state(self, params, obs_names, address):
        """params: reference state_dict names -> arrays (golden files)."""
        def put(layer, prefix):
            layer.weight.data = torch.as_tensor(params[prefix + '.weight'], dtype=torch.float32).clone()
            layer.bias.data = torch.as_tensor(params[prefix + '.bias'], dtype=torch.float32).clone()
        for o, name in enumerate(obs_names):
            for i in range(2):
                put(self.obs[o][i], '_layers_observe_embedding.%s._layers.%d' % (name, i))
        for i in range(2):
            put(self.final[i], '_layers_observe_embedding_final._layers.%d' % i)
            put(self.proposal[i], '_layers_proposal.%s._ff._layers.%d' % (address, i))
        put(self.sample_embedding, '_layers_sample_embedding.%s._layers.0' % address)
        for k in ('weight_ih_l0', 'weight_hh_l0', 'bias_ih_l0', 'bias_hh_l0'):
            getattr(self.lstm, k).data = torch.as_tensor(params['_layers_lstm.' + k], dtype=torch.float32).clone()
        self.address_embedding.data = torch.as_tensor(params['_layers_address_embedding.' + address], dtype=torch.float32).clone()
        dt = [k for k in params if k.startswith('_layers_distribution_type_embedding.')][0]
        self.dtype_embedding.data = torch.as_tensor(params[dt], dtype=torch.float32).clone()

    def loss(self, obs, value, prior_mean, prior_stddev):
        """_loss for a minibatch of single-statement traces: obs [B, n_obs], value / prior_* [B]."""
        B = obs.shape[0]
        parts = []
        for o, ff in enumerate(self.obs):                                               # _embed_observe, :132-139
            parts.append(torch.relu(ff[1](torch.relu(ff[0](obs[:, o:o + 1])))))
        e = torch.cat(parts, 1)
        e = torch.relu(self.final[1](torch.relu(self.final[0](e))))
        x = torch.cat([e, e.new_zeros(B, self.zeros), self.dtype_embedding.expand(B, -1),
                       self.address_embedding.expand(B, -1)], 1).unsqueeze(0)           # lstm.py:175-185
        h0 = e.new_zeros(1, B, self.lstm.hidden_size)
        out, _ = self.lstm(x, (h0, h0.clone()))                                         # :186-188
        y = self.proposal[1](torch.relu(self.proposal[0](out[0])))
        K = self.K
        means = prior_mean[:, None] + y[:, :K] * prior_stddev[:, None]                  # proposal_normal_normal_mixture.py:20-35
        stddevs = torch.exp(y[:, K:2 * K]) * prior_stddev[:, None]
        probs = torch.softmax(y[:, 2 * K:], 1)
        comp = -((value[:, None] - means) ** 2) / (2 * stddevs ** 2) - stddevs.log() - 0.5 * math.log(2 * math.pi)
        p = probs / probs.sum(1, keepdim=True)                                          # mixture.py:14-16
        lp = torch.logsumexp(torch.log(p.clamp(FP32_EPS, 1 - FP32_EPS)) + comp, 1)      # mixture.py:43-44
        return -lp.sum() / B                                                            # lstm.py:218-220


def time_training_steps(lstm_dim, batch, budget_s=10.0, threads=None, seed=0):
    """zero_grad -> loss -> backward -> Adam.step on fresh synthetic GUM minibatches until the budget is used.
    Returns (traces per second, steps, threads)."""
    import time
    if threads:
        torch.set_num_threads(threads)
    torch.manual_seed(seed)
    net = GumNetwork(lstm_dim)
    opt = torch.optim.Adam(net.parameters(), lr=1e-3, weight_decay=0.0)
    g = torch.Generator().manual_seed(seed)

    def minibatch():
        mu = 1.0 + math.sqrt(5.0) * torch.randn(batch, generator=g)
        obs = mu[:, None] + math.sqrt(2.0) * torch.randn(batch, 2, generator=g)
        return obs, mu, torch.full((batch,), 1.0), torch.full((batch,), math.sqrt(5.0))
    for _ in range(2):                        # warm-up: thread pools, allocator
        opt.zero_grad()
        net.loss(*minibatch()).backward()
        opt.step()
    steps, t0 = 0, time.time()
    while True:
        opt.zero_grad()
        loss = net.loss(*minibatch())
        loss.backward()
        opt.step()
        float(loss)                           # inference_network.py:497
        steps += 1
        if steps >= 3 and time.time() - t0 > budget_s:
            break
    dt = time.time() - t0
    return steps * batch / dt, steps, torch.get_num_threads()
