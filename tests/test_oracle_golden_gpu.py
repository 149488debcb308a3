"""The pin of the oracle (tests/test_oracle_golden.py: oracle/knn_exact.c and oracle/torch_ref.py against the fixtures
generated from the imported reference) re-run under `-m gpu`, i.e. ON THE GPU BOX, whose host CPU and C library are not
this container's: the parity tests there trust libknn_exact.so as built and run on that host."""
import pytest

from test_oracle_golden import *  # noqa: F401,F403  (the same test functions, collected a second time under the gpu marker)

pytestmark = pytest.mark.gpu
