"""ref: lxt/explicit/check.py:6-15 -- context manager switching the rule Functions into conservation-check mode"""
from .functional import CONSERVATION_CHECK_FLAG


class conservation_check(object):
    def __enter__(self):
        CONSERVATION_CHECK_FLAG[0] = True

    def __exit__(self, exc_type, exc_value, traceback):
        CONSERVATION_CHECK_FLAG[0] = False
