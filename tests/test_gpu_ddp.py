"""The data-parallel step on ONE GPU with a real RCCL group (world 1, forced): GradBuckets' in-backward gradient sinks,
the bucket all-reduces launched from inside the backward, the per-step buffer broadcast and the whole thing captured into
a hipGraph - against the same step without any of it (tools/ddp_selfcheck.py, run in a child process: the process group's
watchdog thread and destructor stay out of this process, where other tests capture hipGraphs).  A 1-rank AVG all-reduce
is the identity, so gradients and parameters must come out BIT-identical: an aliasing / ordering bug in the sink path
(a column-sum partial landing in the wrong slot, a collective reading a slot before the side-stream weight gradient wrote
it) shows up as a difference.  SURVEY.md 8(e); reference: DistributedDataParallel, engine/processor.py:47-50."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("dtype", ["bf16", "f16", "bf16 wire16"])
def test_grad_bucket_sinks_equal_plain_autograd(dtype):
    import torch
    torch.cuda.empty_cache()           # this process may hold tens of GB of cached blocks from earlier tests: the child needs its own
    out = ""
    for attempt in range(2):           # (one retry for a failed rendezvous.  The intermittent abort this test used to show - 1 run in 6 -
                                       #  was the process group's watchdog querying an event while the capture was open: fixed
                                       #  by editor_amd.ddp.graph_capture_kwargs, thread-local capture mode)
        cp = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ddp_selfcheck.py")] + dtype.split(), stdout=subprocess.PIPE,
                            stderr=subprocess.STDOUT, text=True, timeout=900)
        out = cp.stdout
        if cp.returncode == 0 and "DDP-SELFCHECK-OK" in out:
            return
        if "AssertionError" in out or "differ" in out:
            break                      # a real mismatch is never retried
    raise AssertionError(out[-3000:])
