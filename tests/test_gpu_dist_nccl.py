"""RCCL for real on the one GPU a test box has ("RCCL all-reduce of gradients over xGMI in training",
the fit of given_others_gt_mean_var_seq2seq.py:494-506).  The worker runs as a CHILD process - started before it touches the
GPU - with backend "nccl" (= RCCL on ROCm) at world size 1 and FOV_FORCE_DIST=1, which sends the trainers through their
data-parallel branch: async tail all-reduce under the encoder BPTT, poison slot + head all-reduce, work.wait(), guarded
optimizer; parallel.broadcast_index on the nccl backend."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_forced_data_parallel_over_rccl_equals_plain_path():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("FOV_FORCE_DIST", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dist_nccl_worker.py")], env=env, capture_output=True,
                       text=True, timeout=900)
    print(r.stdout[-4000:])
    print(r.stderr[-4000:])
    assert r.returncode == 0, "worker failed"
    assert "DIST_NCCL_WORKER OK" in r.stdout
    for name in ("mixing_f32", "mixing_bf16", "seq2seq"):
        assert "%s: forced-DP over RCCL == plain path: True" % name in r.stdout
