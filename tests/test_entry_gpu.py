"""GPU: the driver's entry points keep working (smoke() is what runs first on the MI355X box at round end)."""
import pytest

pytestmark = pytest.mark.gpu


def test_smoke_entry_point():
    import __graft_entry__ as g
    g.smoke()
