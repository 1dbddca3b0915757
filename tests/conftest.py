import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")
    # The CPU oracle (the checker of the parity tests) runs through torch's CPU kernels.  On the GPU box's 128-core host torch defaults to 128 intra-op
    # threads, and the oracle's many small / gather-scatter kernels then run SLOWER than on 8 cores: autograd through the oracle took 38 s per backward
    # pass there against 3 s here (cProfile of test_eval_mode_gradients_of_all_four_losses_strict[3], round 6; profiles/r02_cpu_baseline_threads.txt shows
    # the same for the forward: 128 threads 0.035 scans/s, 32 threads 0.116).  32 threads is what bench.py's cpu_baseline uses for the same reason.
    try:
        import torch

        torch.set_num_threads(min(32, os.cpu_count() or 1))
    except Exception:
        pass


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)

    return load


# ---- run-time record of the GPU suite (the driver's step limit is 1 200 s) -------------------------------------------------
# Every `-m gpu` run on a GPU box writes the per-test durations (setup + call + teardown, slowest first) to gpurun_out/gputest_durations.txt; the
# copy judged is profiles/r06_gputest_durations.txt.
_DUR = {}


def pytest_runtest_logreport(report):
    _DUR[report.nodeid] = _DUR.get(report.nodeid, 0.0) + float(getattr(report, "duration", 0.0))


def pytest_sessionfinish(session, exitstatus):
    try:
        import torch

        if not torch.cuda.is_available() or not _DUR:
            return
        out = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        rows = sorted(_DUR.items(), key=lambda kv: -kv[1])
        with open(os.path.join(out, "gputest_durations.txt"), "w") as f:
            f.write(f"# pytest per-test durations (s), {len(rows)} tests, total {sum(v for _, v in rows):.1f} s, exit status {int(exitstatus)}\n")
            for k, v in rows:
                f.write(f"{v:9.2f}  {k}\n")
    except Exception:            # a bookkeeping failure must never fail the suite
        pass
