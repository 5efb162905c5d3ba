"""Import-time stand-in for `pydotplus` (pyprob/graph.py). Graph rendering is out of scope."""
