"""Deterministic synthetic weights / inputs shared by the golden generator and the tests.

The helpers live in the package (``pafuse_amd/synthetic.py``: the benchmark, the driver entry and the examples use
them too and must not depend on the test tree); this module keeps the name the fixtures were generated under.
"""
from pafuse_amd.synthetic import *  # noqa: F401,F403
from pafuse_amd.synthetic import _key_generator  # noqa: F401
