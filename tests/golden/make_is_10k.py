#!/usr/bin/env python3
"""10 000 importance-sampling particles per program, recorded by RUNNING THE REFERENCE with the committed golden networks
(tests/golden/gum_net.npz, gumm_net.npz): SURVEY.md §8(c) asks for the log-weights of 10^4 reference-sampled particles as
the pin of the 1e-4 tolerance BASELINE.json states. Build container only (needs /root/reference):

    python tests/golden/make_is_10k.py

The network is rebuilt with the reference's own classes (same program classes as make_golden.py, so the addresses - which
contain bytecode offsets - are the recorded ones), its layers created by `_polymorph`, the recorded state_dict loaded, and
`posterior`'s trace generator run for 10 000 particles (pyprob/model.py:59-71, state.py:203-219). Per particle the arrays
hold what the reference computed: values, prior parameters, prior log_prob (state.py:211), proposal log_prob (:212), the sum
of the observed likelihood terms (:147-149), the trace log-weight (trace.py:123-125) and the result. Written to
<case>_is10k.npz in the layout of <case>_is.npz (without the proposal parameters)."""
import importlib.util
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(REPO, 'oracle', 'refstubs'))
sys.path.insert(1, '/root/reference')

import numpy as np  # noqa: E402
import torch  # noqa: E402
import pyprob  # noqa: E402
from pyprob import InferenceEngine  # noqa: E402
from pyprob.nn import Batch, InferenceNetworkLSTM  # noqa: E402

spec = importlib.util.spec_from_file_location('make_golden', os.path.join(HERE, 'make_golden.py'))
G = importlib.util.module_from_spec(spec)
spec.loader.exec_module(G)          # (its __main__ block does not run: only the program classes and helpers are used)

N = 10000
OBSERVE = {'obs0': 8, 'obs1': 9}


def network(case, model):
    meta = json.load(open(os.path.join(HERE, case + '_meta.json')))
    npz = np.load(os.path.join(HERE, case + '_net.npz'))
    sd = {n: torch.from_numpy(npz['p%d' % i]) for i, n in enumerate(meta['state_dict_names'])}
    emb = {n: {'dim': meta['observe_embedding_dims'][n]} for n in meta['obs_names']}
    net = InferenceNetworkLSTM(model=model, observe_embeddings=emb, lstm_dim=meta['lstm_dim'])
    pyprob.seed(1)
    gen = model._trace_generator(trace_mode=pyprob.TraceMode.PRIOR_FOR_INFERENCE_NETWORK)
    want = {k[len('_layers_address_embedding.'):] for k in sd if k.startswith('_layers_address_embedding.')}
    traces, seen = [], set()
    while seen != want:                # the layers of the recorded network, created by the reference's own _polymorph
        t = next(gen)
        if all(v.address in want for v in t.variables_controlled):
            traces.append(t)
            seen.update(v.address for v in t.variables_controlled)
    net._init_layers_observe_embedding(emb, example_trace=traces[0])
    net._init_layers()
    net._layers_initialized = True
    net._polymorph(Batch(traces))
    missing = set(sd.keys()) ^ set(net.state_dict().keys())
    assert not missing, missing
    net.load_state_dict(sd)
    net.eval()
    return net, meta


def record(case, model):
    net, meta = network(case, model)
    steps = []
    orig = net._infer_step

    def infer_step(variable, prev_variable=None, proposal_min_train_iterations=None):
        d = orig(variable, prev_variable=prev_variable, proposal_min_train_iterations=proposal_min_train_iterations)
        steps.append((variable.address, d))
        return d
    net._infer_step = infer_step
    pyprob.seed(2024)
    gen = model._trace_generator(trace_mode=pyprob.TraceMode.POSTERIOR,
                                 inference_engine=InferenceEngine.IMPORTANCE_SAMPLING_WITH_INFERENCE_NETWORK,
                                 inference_network=net, observe=OBSERVE)
    rows = dict(trace_len=[], addr=[], value=[], prior=[], prior_lp=[], prop_lp=[], lw=[], obs_lw=[], result=[])
    addresses = list(meta['is_addresses'])
    addresses += sorted(a for a in net._layers_proposal.keys() if a not in addresses)      # (layers the 48 recorded particles never reached)
    with torch.no_grad():
        while len(rows['lw']) < N:
            steps.clear()
            tr = next(gen)
            if any(v.address not in addresses for v in tr.variables_controlled):
                continue          # an iteration count the golden network has no layers for (prior as proposal): not recorded
            rows['trace_len'].append(tr.length_controlled)
            rows['lw'].append(float(tr.log_importance_weight))
            rows['obs_lw'].append(sum(float(v.log_importance_weight) for v in tr.variables_observed))
            rows['result'].append(float(tr.result))
            for v, (addr, d) in zip(tr.variables_controlled, steps):
                assert addr == v.address
                rows['addr'].append(addresses.index(addr))
                rows['value'].append(float(v.value))
                _, pp = G.prior_params(v.distribution)
                rows['prior'].append(pp + [0.0] * (3 - len(pp)))
                rows['prior_lp'].append(float(v.log_prob))
                rows['prop_lp'].append(float(d.log_prob(v.value, sum=True)))
    out = dict(trace_len=np.array(rows['trace_len'], np.int32), addr=np.array(rows['addr'], np.int32),
               value=np.array(rows['value'], np.float32), prior=np.array(rows['prior'], np.float32),
               prior_lp=np.array(rows['prior_lp'], np.float64), prop_lp=np.array(rows['prop_lp'], np.float64),
               lw=np.array(rows['lw'], np.float64), obs_lw=np.array(rows['obs_lw'], np.float64),
               result=np.array(rows['result'], np.float32),
               observe=np.array([float(OBSERVE[n]) for n in meta['obs_names']], np.float32),
               addresses=np.array(addresses))
    np.savez_compressed(os.path.join(HERE, case + '_is10k.npz'), **out)
    lw = out['lw']
    w = np.exp(lw - lw.max())
    print(case, 'particles', len(lw), 'statements', len(out['value']), 'lw range', float(lw.min()), float(lw.max()),
          'ESS', float(w.sum() ** 2 / (w * w).sum()))


if __name__ == '__main__':
    record('gum', G.GaussianWithUnknownMean())
    record('gumm', G.GaussianWithUnknownMeanMarsaglia())
