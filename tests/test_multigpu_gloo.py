"""The N > 1 path on CPU: world size 2 and 3 over gloo (task rule 5)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world,cls", [(2, "A"), (3, "S")])
def test_row_partition_host_gather_over_gloo(world, cls, pkg, oracle):
    port = 29610 + world
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "_gloo_worker.py"), cls, "%d_%d" % (os.getpid(), world)]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, timeout=600)
    assert proc.returncode == 0, proc.stdout[-3000:]
    line = [l for l in proc.stdout.splitlines() if l.startswith("GLOO_RESULT")]
    assert line and line[0].split()[1] == "OK", proc.stdout[-3000:]
