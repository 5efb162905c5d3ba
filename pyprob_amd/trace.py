"""Variable / Trace records of the trace runtime (mirror of pyprob/trace.py:9-199 for the fields the hot path reads)."""
import torch


class Variable:
    def __init__(self, distribution=None, value=None, address_base=None, address=None, instance=None, log_prob=None,
                 log_importance_weight=None, control=False, name=None, observed=False, reused=False, tagged=False):
        self.distribution = distribution
        self.value = value if (value is None or torch.is_tensor(value)) else torch.as_tensor(value, dtype=torch.float32)
        self.address_base = address_base
        self.address = address
        self.instance = instance
        self.log_prob = log_prob
        self.log_importance_weight = log_importance_weight
        self.control = control
        self.name = name
        self.observable = (not tagged) and (name is not None)
        self.observed = observed
        self.reused = reused
        self.tagged = tagged

    def __repr__(self):
        return 'Variable(name:{}, control:{}, observed:{}, address:{}, value:{})'.format(
            self.name, self.control, self.observed, self.address, self.value)


class Trace:
    def __init__(self):
        self.variables = []
        self.variables_controlled = []
        self.variables_observed = []
        self.variables_dict_address_base = {}
        self.named_variables = {}
        self.result = None
        self.log_prob = 0.
        self.log_prob_observed = 0.
        self.log_importance_weight = 0.
        self.length = 0
        self.length_controlled = 0
        self.execution_time_sec = None

    def add(self, variable):
        self.variables.append(variable)
        self.variables_dict_address_base[variable.address_base] = variable

    def last_instance(self, address_base):
        v = self.variables_dict_address_base.get(address_base)
        return 0 if v is None else v.instance

    def end(self, result, execution_time_sec):
        """pyprob/trace.py:106-125: collect controlled/observed variables and sum the log-importance-weights in
        Python double precision."""
        self.result = result
        self.execution_time_sec = execution_time_sec
        for v in self.variables:
            if v.name is not None:
                self.named_variables[v.name] = v
            if v.control:
                self.variables_controlled.append(v)
        self.variables_observed = [v for v in self.variables if v.observed]
        self.log_prob = sum(float(torch.sum(v.log_prob)) for v in self.variables
                            if (v.control or v.observed) and v.log_prob is not None)
        # (log_prob is None for terms that were scored in batch on the device: lock-step / coroutine runs)
        self.log_prob_observed = sum(float(torch.sum(v.log_prob)) for v in self.variables_observed if v.log_prob is not None)
        self.length = len(self.variables)
        self.length_controlled = len(self.variables_controlled)
        for v in self.variables:
            if v.log_importance_weight is not None:
                self.log_importance_weight += v.log_importance_weight

    def __getitem__(self, variable_name):
        """Value of a named variable (pyprob/trace.py:192-196)."""
        if variable_name in self.named_variables:
            return self.named_variables[variable_name].value
        raise RuntimeError('Trace does not include variable with name: {}'.format(variable_name))

    def __contains__(self, variable_name):
        return variable_name in self.named_variables

    def __repr__(self):
        return 'Trace(variables:{}, controlled:{}, log_importance_weight:{})'.format(
            self.length, self.length_controlled, self.log_importance_weight)
