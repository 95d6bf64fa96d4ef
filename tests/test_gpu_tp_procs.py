"""Tensor parallelism across two real PROCESSES sharing card 0 (the multi-rank code path that
one GPU allows): see tests/tp_proc_worker.py."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port() -> int:
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_process_tensor_parallel_engines(tmp_path):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "tp_proc_worker.py"), str(tmp_path)]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    res = [json.load(open(tmp_path / f"rank{i}.json")) for i in range(2)]
    print("RCCL with two ranks on one device:", res[0]["rccl_world2"])
    # the gloo-backed collective: both ranks emit the same streams, speculation is lossless
    assert res[0]["gloo_ar"] == res[1]["gloo_ar"] and res[0]["gloo_spec"] == res[1]["gloo_spec"]
    assert res[0]["gloo_spec"] == res[0]["gloo_ar"]
    # sharding only changes rounding: most tokens equal the unsharded model's
    same = sum(a == b for x, y in zip(res[0]["gloo_ar"], res[0]["unsharded_ar"]) for a, b in zip(x, y))
    assert same >= 0.5 * sum(len(x) for x in res[0]["gloo_ar"])
    # the one-shot IPC collectives: real cross-process exchanges inside kernels
    for rr in res:
        assert rr["ipc"] == "ran", rr["ipc"]
        assert rr["ipc_allreduce_ok"] and rr["ipc_allgather_ok"] and rr["ipc_timeouts"] == 0
    for graph in ("False", "True"):
        assert res[0][f"ipc_stream_graph_{graph}"] == res[1][f"ipc_stream_graph_{graph}"]
        assert res[0][f"ipc_stream_graph_{graph}"] == res[0][f"ipc_ar_graph_{graph}"]   # lossless
        # two ranks: the IPC sum (f32 in rank order, rounded once) is the gloo path's sum
        assert res[0][f"ipc_stream_graph_{graph}"] == res[0]["gloo_spec"]
    assert res[0]["ipc_worker_stream"] == res[1]["ipc_worker_stream"] == res[0]["ipc_stream_graph_True"]
    assert res[0]["ipc_worker_timeouts"] == 0 and res[1]["ipc_worker_timeouts"] == 0
    if res[0]["rccl_world2"] == "communicator created":
        for rr in res:
            assert rr["rccl_allreduce_ok"] and rr["rccl_allgather_ok"]
        for key in ("rccl_stream_graph_False", "rccl_stream_graph_True"):
            assert res[0][key] == res[1][key] == res[0]["gloo_spec"], key
    else:
        assert res[0]["rccl_world2"].startswith("refused") and res[1]["rccl_world2"].startswith("refused")
