"""the unmodified reference trackers on the MI355X through `pytracking_amd.install()` (see
tests/trackers_on_device.py for what is run and compared).  Needs the recipe-built reference bundle oracle/_ref/reference
(oracle/make_ref_bundle.py; `__graft_entry__.build()` writes it wherever /root/reference is mounted) -- skipped without it."""
import pytest

from oracle import ref_harness

import trackers_on_device as TOD

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not ref_harness.available(), reason="no reference tree / bundle (oracle/_ref)")]


@pytest.mark.parametrize("which", ["dimp", "prdimp", "tomp", "atom", "lwl"])
def test_reference_tracker_on_device_matches_its_cpu_run(which):
    dev, stats, times, extra = TOD.check(which, atol=1e-4)
    print(f"{which}: max deviation per boundary payload", {k: f"{v:.2e}" for k, v in sorted(dev.items())})
    print(f"{which}: install.stats", stats, extra)
