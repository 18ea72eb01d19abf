import os
import sys
import threading
import time

import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session', autouse=True)
def _heartbeat():
    """The full-length parity tests run a float64 numpy oracle for minutes without printing; a line every 30 s in
    gpurun_out/ (when that directory exists: the GPU box) tells the harness the run is alive."""
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
    stop = threading.Event()
    if os.path.isdir(d):
        def beat():
            t0 = time.time()
            while not stop.wait(30.0):
                try:
                    with open(os.path.join(d, 'pytest_heartbeat.log'), 'a') as f:
                        f.write('%d s\n' % (time.time() - t0))
                except OSError:
                    pass
        threading.Thread(target=beat, daemon=True).start()
    yield
    stop.set()
