"""Import-time stand-in for `termcolor` (absent in this image; pyprob/util.py:8 needs it).
Test infrastructure only -- used when importing the read-only reference to make golden vectors."""


def colored(s, *args, **kwargs):
    return s
