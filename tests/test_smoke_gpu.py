import pytest


@pytest.mark.gpu
def test_graft_smoke():
    import __graft_entry__
    __graft_entry__.smoke()
