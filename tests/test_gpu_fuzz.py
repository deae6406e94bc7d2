"""A slice of the differential fuzz of the step forms (tools/fuzz_forms.py): random spaces, populations, models, call counts and
pool-step knobs; every case runs the same seeded search in two step forms and compares trees, counters, argmin, improvement
counts and state vectors, across an epoch boundary."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_forty_random_cases_of_the_step_forms_agree():
    import azdopt_amd
    assert azdopt_amd.device_count() > 0, "no MI355X visible"
    import fuzz_forms
    fuzz_forms.run(40, 4)
