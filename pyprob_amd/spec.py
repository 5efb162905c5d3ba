"""Network description and flat parameter layout of the LSTM inference network (and of the FeedForward variant,
pyprob/nn/inference_network_feedforward.py: `network='feedforward'`, no LSTM and no address / sample embeddings - the
proposal layers read the observe embedding).

Tensor names and shapes are exactly the reference's `state_dict` (pyprob/nn/inference_network_lstm.py:29-80,
pyprob/nn/inference_network.py:80-130, pyprob/nn/embedding_feedforward.py:22-33), so that
checkpoints interchange. Every tensor lives in ONE flat fp32 buffer (padded to 1024-float chunks) so that the
gradient all-reduce and the Adam update are single kernels over contiguous HBM.
"""
import ctypes as C
import math
from collections import OrderedDict

import numpy as np

from . import lib as L

CHUNK = 1024

DIST_KIND = {'Normal': L.PP_HEAD_NORMAL_MIXTURE, 'Uniform': L.PP_HEAD_TRUNCNORMAL_MIXTURE,
             'Categorical': L.PP_HEAD_CATEGORICAL, 'Poisson': L.PP_HEAD_POISSON_TN_MIXTURE,
             'Bernoulli': L.PP_HEAD_BERNOULLI}


class AddressInfo:
    def __init__(self, address, dist_name, num_categories=None):
        if dist_name not in DIST_KIND:
            # mirrors inference_network_lstm.py:68 for the distributions this engine covers
            raise RuntimeError('Distribution currently unsupported: {}'.format(dist_name))
        self.address = address
        self.dist_name = dist_name
        self.kind = DIST_KIND[dist_name]
        self.num_categories = num_categories
        self.total_train_iterations = 0   # proposal_layer._total_train_iterations, inference_network_lstm.py:198


class NetSpec:
    """Dimensions + ordered tensor table. Grows when `_polymorph` meets a new address."""

    def __init__(self, observe_embeddings, lstm_dim=512, sample_embedding_dim=4, address_embedding_dim=64,
                 distribution_type_embedding_dim=8, proposal_mixture_components=10, network='lstm', lstm_depth=1):
        if network not in ('lstm', 'feedforward'):
            raise ValueError('network must be lstm or feedforward')
        if not 1 <= int(lstm_depth) <= L.PP_MAX_LSTM_DEPTH:
            raise ValueError('lstm_depth must be 1..%d' % L.PP_MAX_LSTM_DEPTH)
        self.lstm_depth = 1 if network == 'feedforward' else int(lstm_depth)
        self.network = network
        self.feedforward = network == 'feedforward'
        # observe_embeddings: ordered {name: {'dim': D, 'input_dim': d_in}}  (FEEDFORWARD, depth 2 only)
        self.obs = []
        self.obs_depth = {}
        for name, v in observe_embeddings.items():
            depth = int(v.get('depth', 2))          # default 2, inference_network.py:116
            if not 1 <= depth <= L.PP_MAX_OBS_DEPTH:
                raise ValueError('observe embedding depth must be 1..%d' % L.PP_MAX_OBS_DEPTH)
            d_in = int(v.get('input_dim', 1))
            d_out = int(v.get('dim', 256))          # default 256, inference_network.py:103
            self.obs.append((name, d_in, int((d_in + d_out) / 2), d_out))
            self.obs_depth[name] = depth
        if not self.obs:
            raise ValueError('At least one observe embedding is needed to initialize inference network.')
        if len(self.obs) > L.PP_MAX_OBS:
            raise ValueError('at most %d observables' % L.PP_MAX_OBS)
        self.e_obs = sum(o[3] for o in self.obs)
        self.obs_width = sum(o[1] for o in self.obs)
        self.smp_dim = sample_embedding_dim
        self.addr_dim = address_embedding_dim
        self.dtype_dim = distribution_type_embedding_dim
        self.lstm_dim = 0 if self.feedforward else lstm_dim
        self.head_in = self.e_obs if self.feedforward else lstm_dim       # input width of the proposal layers
        self.K = proposal_mixture_components
        self.lstm_in = 0 if self.feedforward else self.e_obs + self.smp_dim + 2 * (self.addr_dim + self.dtype_dim)
        self.addresses = []          # AddressInfo, index = address id
        self.address_id = {}
        self.dtypes = []             # distribution type names, index = dtype id
        self.tensors = OrderedDict()  # name -> (offset, shape)
        self.n_params = 0
        H, I, e = lstm_dim, self.lstm_in, self.e_obs
        for name, d_in, hid, d_out in self.obs:
            p = '_layers_observe_embedding.%s._layers.' % name
            for l, (rows, cols) in enumerate(self.obs_layer_shapes(name)):      # embedding_feedforward.py:22-33
                self._add(p + '%d.weight' % l, (rows, cols)); self._add(p + '%d.bias' % l, (rows,))
        p = '_layers_observe_embedding_final._layers.'
        self._add(p + '0.weight', (e, e)); self._add(p + '0.bias', (e,))
        self._add(p + '1.weight', (e, e)); self._add(p + '1.bias', (e,))
        if not self.feedforward:
            for k in range(self.lstm_depth):     # nn.LSTM parameter order: per layer w_ih, w_hh, b_ih, b_hh
                self._add('_layers_lstm.weight_ih_l%d' % k, (4 * H, I if k == 0 else H))
                self._add('_layers_lstm.weight_hh_l%d' % k, (4 * H, H))
                self._add('_layers_lstm.bias_ih_l%d' % k, (4 * H,)); self._add('_layers_lstm.bias_hh_l%d' % k, (4 * H,))
        self.n_core_tensors = len(self.tensors)

    def obs_layer_shapes(self, name):
        """[(out, in)] of the Linear layers of observable `name` (EmbeddingFeedForward(num_layers = depth))."""
        _, d_in, hid, d_out = [o for o in self.obs if o[0] == name][0]
        depth = self.obs_depth[name]
        if depth == 1:
            return [(d_out, d_in)]
        return [(hid, d_in)] + [(hid, hid)] * (depth - 2) + [(d_out, hid)]

    # ---- layout ------------------------------------------------------------------------------------
    def _add(self, name, shape):
        n = int(np.prod(shape))
        self.tensors[name] = (self.n_params, tuple(shape))
        self.n_params += ((n + CHUNK - 1) // CHUNK) * CHUNK

    def offset(self, name):
        return self.tensors[name][0]

    def tensor_index(self, name):
        return list(self.tensors.keys()).index(name)

    @property
    def n_tensors(self):
        return len(self.tensors)

    def num_parameters(self):
        """Unpadded parameter count (the number pyprob prints, inference_network_lstm.py:76-77)."""
        return int(sum(int(np.prod(s)) for _, s in self.tensors.values()))

    def head_dims(self, info):
        n_out = (info.num_categories if info.kind == L.PP_HEAD_CATEGORICAL else
                 1 if info.kind == L.PP_HEAD_BERNOULLI else 3 * self.K)     # proposal_bernoulli_bernoulli.py:13
        hid = int((self.head_in + n_out) / 2)     # embedding_feedforward.py:26
        smp_in = info.num_categories if info.kind == L.PP_HEAD_CATEGORICAL else 1
        return n_out, hid, smp_in

    def add_address(self, address, dist_name, num_categories=None):
        """New layers for a new address (`_polymorph`, inference_network_lstm.py:42-73). Returns the names of
        the tensors that were created."""
        if address in self.address_id:
            return []
        info = AddressInfo(address, dist_name, num_categories)
        created = []
        n0 = len(self.tensors)
        if not self.feedforward:
            self._add('_layers_address_embedding.' + address, (self.addr_dim,))
        if dist_name not in self.dtypes:
            self.dtypes.append(dist_name)
            if not self.feedforward:
                self._add('_layers_distribution_type_embedding.' + dist_name, (self.dtype_dim,))
        n_out, hid, smp_in = self.head_dims(info)
        p = '_layers_proposal.%s._ff._layers.' % address
        self._add(p + '0.weight', (hid, self.head_in)); self._add(p + '0.bias', (hid,))
        self._add(p + '1.weight', (n_out, hid)); self._add(p + '1.bias', (n_out,))
        if not self.feedforward:
            p = '_layers_sample_embedding.%s._layers.0.' % address
            self._add(p + 'weight', (self.smp_dim, smp_in)); self._add(p + 'bias', (self.smp_dim,))
        created = list(self.tensors.keys())[n0:]
        self.address_id[address] = len(self.addresses)
        self.addresses.append(info)
        return created

    # ---- initial values (PyTorch defaults the reference relies on) --------------------------------------
    def init_tensor(self, name, rng):
        """nn.Linear / nn.LSTM default init (U(-1/sqrt(fan), 1/sqrt(fan))) and N(0,1) embeddings
        (inference_network_lstm.py:43,47)."""
        _, shape = self.tensors[name]
        if name.startswith('_layers_address_embedding.') or name.startswith('_layers_distribution_type_embedding.'):
            return rng.standard_normal(shape).astype(np.float32)
        if name.startswith('_layers_lstm.'):
            k = 1.0 / math.sqrt(self.lstm_dim)
            return rng.uniform(-k, k, shape).astype(np.float32)
        # Linear: fan_in = weight.shape[1]; the bias uses the same bound
        if name.endswith('.weight'):
            fan_in = shape[1]
        else:
            fan_in = self.tensors[name[:-len('bias')] + 'weight'][1][1]
        k = 1.0 / math.sqrt(fan_in)
        return rng.uniform(-k, k, shape).astype(np.float32)

    # ---- which tensors take part in a batch (grad is not None in the reference) ---------------------------
    def active_mask(self, cur_counts, prev_counts):
        """float32 [n_tensors]: 1 where the reference's autograd would produce a gradient for this batch:
        core layers always; per address: head + embeddings if it occurs, sample embedding only if it occurs as a
        PREVIOUS variable (inference_network_lstm.py:168-171)."""
        act = np.zeros(self.n_tensors, np.float32)
        act[:self.n_core_tensors] = 1.0
        names = list(self.tensors.keys())
        index = {n: i for i, n in enumerate(names)}
        for a, info in enumerate(self.addresses):
            cur = cur_counts[a] > 0
            prev = prev_counts[a] > 0
            if (cur or prev) and not self.feedforward:
                act[index['_layers_address_embedding.' + info.address]] = 1.0
                act[index['_layers_distribution_type_embedding.' + info.dist_name]] = 1.0
            if cur:
                for s in ('0.weight', '0.bias', '1.weight', '1.bias'):
                    act[index['_layers_proposal.%s._ff._layers.%s' % (info.address, s)]] = 1.0
            if prev and not self.feedforward:
                for s in ('weight', 'bias'):
                    act[index['_layers_sample_embedding.%s._layers.0.%s' % (info.address, s)]] = 1.0
        return act

    def tensor_roles(self):
        """active_mask as tables for the native training loop (pp_tensor_roles, include/pyprob_amd.h): per tensor the
        addresses whose occurrence gives it a gradient and how (bit 0: as current variable, bit 1: as previous variable,
        bit 2: always). Returns int32 arrays (off [n_tensors+1], addr, role [n_tensors])."""
        off, addr, role = [0], [], []
        by_dtype = {}
        for a, info in enumerate(self.addresses):
            by_dtype.setdefault(info.dist_name, []).append(a)
        for i, name in enumerate(self.tensors):
            if i < self.n_core_tensors:
                role.append(4)
            elif name.startswith('_layers_address_embedding.'):
                addr.append(self.address_id[name[len('_layers_address_embedding.'):]])
                role.append(3)
            elif name.startswith('_layers_distribution_type_embedding.'):
                addr.extend(by_dtype[name[len('_layers_distribution_type_embedding.'):]])
                role.append(3)
            elif name.startswith('_layers_proposal.'):
                addr.append(self.address_id[name[len('_layers_proposal.'):name.index('._ff._layers.')]])
                role.append(1)
            elif name.startswith('_layers_sample_embedding.'):
                addr.append(self.address_id[name[len('_layers_sample_embedding.'):name.index('._layers.0.')]])
                role.append(2)
            else:
                raise RuntimeError('tensor without a role: ' + name)
            off.append(len(addr))
        return np.asarray(off, np.int32), np.asarray(addr + [0], np.int32), np.asarray(role, np.int32)

    def chunk_tensor_map(self):
        m = np.empty(self.n_params // CHUNK, np.int32)
        offs = [o for o, _ in self.tensors.values()] + [self.n_params]
        for t in range(self.n_tensors):
            m[offs[t] // CHUNK:offs[t + 1] // CHUNK] = t
        return m

    # ---- C structs ---------------------------------------------------------------------------------
    def address_table(self):
        """int64 [n_addr, 8] rows for per-row dispatch on the device (PP_AT_* columns)."""
        t = np.zeros((max(len(self.addresses), 1), L.PP_ADDR_TABLE_COLS), np.int64)
        for a, info in enumerate(self.addresses):
            n_out, hid, smp_in = self.head_dims(info)
            if self.feedforward:          # no embeddings: the table only carries the head kind / width
                t[a] = [info.kind, smp_in, 0, 0, 0, 0, n_out, 0]
                continue
            t[a] = [info.kind, smp_in, self.offset('_layers_address_embedding.' + info.address),
                    self.offset('_layers_distribution_type_embedding.' + info.dist_name),
                    self.offset('_layers_sample_embedding.%s._layers.0.weight' % info.address),
                    self.offset('_layers_sample_embedding.%s._layers.0.bias' % info.address), n_out, 0]
        return t

    def c_struct(self, addr_table_dev_ptr):
        net = L.pp_net()
        net.n_obs = len(self.obs)
        for o, (name, d_in, hid, d_out) in enumerate(self.obs):
            p = '_layers_observe_embedding.%s._layers.' % name
            net.obs_in[o], net.obs_hid[o], net.obs_out[o] = d_in, hid, d_out
            depth = self.obs_depth[name]
            net.obs_depth[o] = depth
            for l in range(depth):
                net.obs_w[o][l], net.obs_b[o][l] = self.offset(p + '%d.weight' % l), self.offset(p + '%d.bias' % l)
            if depth == 2:
                net.obs_w0[o], net.obs_b0[o] = self.offset(p + '0.weight'), self.offset(p + '0.bias')
                net.obs_w1[o], net.obs_b1[o] = self.offset(p + '1.weight'), self.offset(p + '1.bias')
        net.e_obs, net.smp_dim, net.addr_dim, net.dtype_dim = self.e_obs, self.smp_dim, self.addr_dim, self.dtype_dim
        p = '_layers_observe_embedding_final._layers.'
        net.fin_w0, net.fin_b0 = self.offset(p + '0.weight'), self.offset(p + '0.bias')
        net.fin_w1, net.fin_b1 = self.offset(p + '1.weight'), self.offset(p + '1.bias')
        net.lstm_in, net.lstm_dim = self.lstm_in, self.lstm_dim          # lstm_dim == 0: FeedForward network
        if not self.feedforward:
            net.w_ih, net.w_hh = self.offset('_layers_lstm.weight_ih_l0'), self.offset('_layers_lstm.weight_hh_l0')
            net.b_ih, net.b_hh = self.offset('_layers_lstm.bias_ih_l0'), self.offset('_layers_lstm.bias_hh_l0')
            net.lstm_depth = self.lstm_depth
            for k in range(self.lstm_depth):
                net.lstm_w_ih[k] = self.offset('_layers_lstm.weight_ih_l%d' % k)
                net.lstm_w_hh[k] = self.offset('_layers_lstm.weight_hh_l%d' % k)
                net.lstm_b_ih[k] = self.offset('_layers_lstm.bias_ih_l%d' % k)
                net.lstm_b_hh[k] = self.offset('_layers_lstm.bias_hh_l%d' % k)
        net.n_addr, net.n_dtype = len(self.addresses), len(self.dtypes)
        arr = (L.pp_addr * max(len(self.addresses), 1))()
        for a, info in enumerate(self.addresses):
            n_out, hid, smp_in = self.head_dims(info)
            r = arr[a]
            r.kind, r.n_out, r.hid, r.smp_in = info.kind, n_out, hid, smp_in
            r.dtype_id = self.dtypes.index(info.dist_name)
            if not self.feedforward:
                r.addr_emb = self.offset('_layers_address_embedding.' + info.address)
                r.dtype_emb = self.offset('_layers_distribution_type_embedding.' + info.dist_name)
                r.smp_w = self.offset('_layers_sample_embedding.%s._layers.0.weight' % info.address)
                r.smp_b = self.offset('_layers_sample_embedding.%s._layers.0.bias' % info.address)
            p = '_layers_proposal.%s._ff._layers.' % info.address
            r.w1, r.b1 = self.offset(p + '0.weight'), self.offset(p + '0.bias')
            r.w2, r.b2 = self.offset(p + '1.weight'), self.offset(p + '1.bias')
        net.addrs = C.cast(arr, C.POINTER(L.pp_addr))
        net.addr_table = addr_table_dev_ptr
        net.n_params = self.n_params
        net._keep = arr  # keep the host array alive
        return net
