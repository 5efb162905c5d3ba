"""Import-time stand-in for `pyzmq` (pyprob/remote.py:2). The PPX remote path is out of scope."""
REQ = 0


class Context:
    def __init__(self, *a, **k):
        raise RuntimeError('zmq stub: remote models are out of scope')
