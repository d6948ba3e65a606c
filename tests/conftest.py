import os
import sys

import pytest
import torch  # noqa: F401  -- before libdalign.so: its bundled HIP runtime must be the one the library binds to

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
  if p not in sys.path:
    sys.path.insert(0, p)


def pytest_configure(config):
  config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
  config.addinivalue_line("markers", "slow: long-running CPU test")
