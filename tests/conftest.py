import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


@pytest.fixture
def tmp_path(tmp_path):
    """RAM-backed scratch directory when the box has one: the checkpoint tests write ~1 GB files and the container's
    /tmp disk sustains ~20 MB/s."""
    import pathlib
    import shutil
    import tempfile
    if os.path.isdir('/dev/shm') and os.access('/dev/shm', os.W_OK):
        d = tempfile.mkdtemp(prefix='es_test_', dir='/dev/shm')
        try:
            yield pathlib.Path(d)
        finally:
            shutil.rmtree(d, ignore_errors=True)
    else:
        yield tmp_path
