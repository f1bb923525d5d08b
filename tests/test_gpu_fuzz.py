"""GPU: a slice of the extended fuzzers (scripts/fuzz_lru_forms.py, scripts/fuzz_ingest.py) in the
suite: the LRU replacement order in both forms and the ingest in random chunk sizes on both
sides of the device-ordering threshold, against the oracle after every step."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))


@pytest.mark.parametrize("form", ["list", "queue"])
@pytest.mark.parametrize("seed", range(1000, 1012))
def test_lru_order_fuzz(seed, form, monkeypatch):
    import fuzz_lru_forms
    monkeypatch.delenv("GNNFLOW_LRU_QUEUE_MIN_CAPACITY", raising=False)
    fuzz_lru_forms.run(seed, 40, form)
    os.environ.pop("GNNFLOW_LRU_QUEUE_MIN_CAPACITY", None)


@pytest.mark.parametrize("seed", range(2000, 2016))
def test_ingest_fuzz(seed):
    import fuzz_ingest
    fuzz_ingest.run(seed)
