"""The one collective of the path on the hardware that is there: RCCL with a group of ONE rank on the one-GPU box (VERDICT r5,
missing #2 / next #3 -- five rounds of world-size-2 gloo tests had never loaded RCCL).  A broadcast in a group of one moves no
bytes over xGMI, but it does everything else the 8-GPU run does: the communicator is created on the device, the header and the
payload are device buffers, the collective is a kernel on RCCL's stream, `broadcast_prompt` waits for it against its deadline and
orders the current stream after it.  Also here, on CPU: the deadline wait itself against a work object that never completes
(what a broadcast whose peer died looks like on RCCL, where `Work.wait(timeout)` only fences the stream)."""
import os
import subprocess
import sys
import time

import pytest
import torch

from videosd_amd.dispatch import _wait_with_deadline, spawn_workers

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKE = "tests.helpers_fake_pipeline:SessionFakePipeline"


class _Work:
    """stands in for torch.distributed's Work: completes after `after` seconds (never: None), optionally failing"""

    def __init__(self, after=None, fail_query=False, fail_wait=False):
        self.t0, self.after, self.fail_query, self.fail_wait, self.waited = time.monotonic(), after, fail_query, fail_wait, 0

    def is_completed(self):
        if self.fail_query:
            raise RuntimeError("NCCL communicator was aborted")
        return self.after is not None and time.monotonic() - self.t0 >= self.after

    def wait(self, *a):
        assert not a, "the deadline is the host clock's: wait() must be called without a timeout, after completion"
        self.waited += 1
        if self.fail_wait:
            raise RuntimeError("connection closed by peer")
        return True


def test_deadline_wait_raises_on_a_collective_that_never_completes():
    w = _Work(after=None)
    t0 = time.monotonic()
    with pytest.raises(RuntimeError, match="timed out after 0.3"):
        _wait_with_deadline(w, 0.3)
    assert 0.3 <= time.monotonic() - t0 < 1.0 and w.waited == 0  # (never fenced the stream behind a collective that is stuck)
    w = _Work(after=0.05)
    _wait_with_deadline(w, 2.0)
    assert w.waited == 1
    with pytest.raises(RuntimeError, match="failed: NCCL communicator was aborted"):
        _wait_with_deadline(_Work(fail_query=True), 1.0)
    with pytest.raises(RuntimeError, match="failed: connection closed"):
        _wait_with_deadline(_Work(after=0.0, fail_wait=True), 1.0)


RANK = r"""
import os, sys, json, time
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
from videosd_amd.dispatch import broadcast_prompt, PROMPT_HEADER_KEYS
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[2])
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
emb = (torch.randn(77, 768, generator=torch.Generator().manual_seed(5)) * 0.7).half()
hdr = {"epoch": 7, "height": 512, "width": 512, "steps": 4, "strength": 0.6, "controlnet_scale": 1.25, "seed": 23}
out = {}
for name, to in (("deadline", 3.0), ("plain", None)):
    t0 = time.time()
    buf, got = broadcast_prompt(emb, hdr, src=0, device=dev, timeout=to)
    torch.cuda.current_stream().synchronize()
    out[name] = dict(seconds=time.time() - t0, device=str(buf.device), equal=bool(torch.equal(buf.cpu(), emb)),
                     header_equal=all(got[k] == float(hdr[k]) for k in PROMPT_HEADER_KEYS))
out["backend"] = dist.get_backend()
x = torch.ones(1 << 20, device=dev)
dist.all_reduce(x)                      # (a second kind of collective through the same communicator)
out["all_reduce"] = float(x.sum())
pg = dist.distributed_c10d._get_default_group()
be = pg._get_backend(dev)
out["has_abort"] = hasattr(be, "abort")
dist.destroy_process_group()
print("RESULT " + json.dumps(out))
"""


@pytest.mark.gpu
def test_rccl_group_of_one_broadcasts_the_prompt_on_the_device():
    from videosd_amd.dispatch import free_port

    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-c", RANK, ROOT, str(free_port())], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    import json

    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])
    assert out["backend"] == "nccl"
    for name in ("deadline", "plain"):
        r = out[name]
        assert r["equal"] and r["header_equal"] and r["device"].startswith("cuda"), out
    assert out["deadline"]["seconds"] < 3.0 + 60.0  # (the first collective creates the communicator)
    assert out["all_reduce"] == float(1 << 20)
    assert out["has_abort"], "this torch's ProcessGroupNCCL has no abort(): abandon_group could not tear a stuck broadcast down"


@pytest.mark.gpu
def test_a_worker_in_an_rccl_group_of_one_syncs_a_prompt_and_serves_a_frame():
    """spawn_workers(1, backend="nccl", collective_at_world_1=True): the worker process forms the RCCL group on its GPU and its
    prompt sync goes through `broadcast_prompt` on device buffers with the sync deadline (the path of every worker of the 8-GPU
    node); a frame is served afterwards, and the group is abandoned (communicator abort) without taking the worker down."""
    import numpy as np
    from PIL import Image

    ws = spawn_workers(1, factory=FAKE, backend="nccl", collective_at_world_1=True, model="m", controlnet="c", call_timeout=120.0,
                       sync_timeout=5.0, group_timeout=60.0)
    try:
        rep = ws[0].sync_prompt.remote("a red fox", {"height": 12}).result(timeout=120)
        assert rep["via"] == "nccl" and rep["epoch"] == 1, rep
        st = ws[0].method("prompt_state")()
        assert st["encodes"] == 1 and abs(st["checksum"] - rep["checksum"]) < 1e-3, (st, rep)
        img = Image.fromarray(np.full((12, 16, 3), 20, dtype=np.uint8), "RGB")
        out = ws[0].infer(img, prompt="a red fox", height=12, width=16)
        assert out.size == (16, 12)
        rep2 = ws[0].sync_prompt.remote("a blue whale", {"height": 12}).result(timeout=60)
        assert rep2["via"] == "nccl" and rep2["epoch"] == 2
    finally:
        for w in ws:
            w.close()
