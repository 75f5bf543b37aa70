"""In-library RCCL collectives (ss_comm_*, ss_pr_exchange) on the one-GPU box: a world-1 communicator runs every entry point
(RCCL refuses two ranks on one device, so the multi-rank data path is exercised by bench.py --gpus N on a multi-GPU node);
the state machine and the argument checks are covered here."""
import numpy as np
import pytest

from spaghettisearch_amd import SpaghettiError, engine, synth

pytestmark = pytest.mark.gpu


@pytest.fixture()
def ctx1():
    ctx = engine.Context(0)
    yield ctx
    ctx.close()


def test_world1_communicator_round_trip(ctx1):
    import torch
    assert ctx1.comm_info() == (-1, 0)
    with pytest.raises(SpaghettiError) as ei:            # no communicator yet
        ctx1.comm_allreduce_u64(np.arange(4, dtype=np.uint64))
    assert ei.value.code == 6
    uid = engine.Context.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    ctx1.comm_init(uid, 0, 1)
    assert ctx1.comm_info() == (0, 1)
    with pytest.raises(SpaghettiError) as ei:            # one communicator per context
        ctx1.comm_init(uid, 0, 1)
    assert ei.value.code == 6
    # all-reduce(sum) over one rank = identity: host buffer and device buffer
    h = np.array([0, 1, 2 ** 40 + 7, 2 ** 63 + 5], dtype=np.uint64)
    want = h.copy()
    ctx1.comm_allreduce_u64(h)
    assert np.array_equal(h, want)
    d = torch.arange(1000, dtype=torch.int64, device="cuda")
    ctx1.comm_allreduce_u64(d)
    ctx1.synchronize()
    assert torch.equal(d.cpu(), torch.arange(1000, dtype=torch.int64))
    # all-gather: host and device buffers
    send = np.frombuffer(np.random.default_rng(1).bytes(4096), dtype=np.uint8).copy()
    recv = np.zeros_like(send)
    ctx1.comm_allgather(send, recv, send.nbytes)
    assert np.array_equal(send, recv)
    ds = torch.from_numpy(send).cuda()
    dr = torch.zeros_like(ds)
    ctx1.comm_allgather(ds, dr, send.nbytes)
    ctx1.synchronize()
    assert torch.equal(ds, dr)
    # 2-D decomposition: a world of one splits into one group of one; collectives keep working on the group's communicator
    with pytest.raises(SpaghettiError):
        ctx1.comm_split(-1, 0)
    ctx1.comm_split(0, 0)
    assert ctx1.comm_info() == (0, 1)
    with pytest.raises(SpaghettiError) as ei:            # one split per context
        ctx1.comm_split(0, 0)
    assert ei.value.code == 6
    h2 = np.arange(5, dtype=np.uint64)
    ctx1.comm_allreduce_u64(h2)
    assert np.array_equal(h2, np.arange(5, dtype=np.uint64))
    ctx1.comm_destroy()
    assert ctx1.comm_info() == (-1, 0)
    ctx1.comm_destroy()                                   # idempotent


def test_exchange_state_machine(ctx1):
    n, e = 3000, 12000
    ptr, dst = synth.rmat_graph(n, e, seed=9)
    n_topic = synth.topic_sizes(n, 2)
    g1 = engine.Graph(ctx1, n, ptr, dst)                              # unsharded graph: no exchange exists
    st = engine.PageRankState(g1, 0.75, 1e-9, n_topic)
    st.begin()
    with pytest.raises(SpaghettiError) as ei:
        st.exchange()
    assert ei.value.code == 6
    with pytest.raises(SpaghettiError) as ei:
        g1.pagerank_sharded(0.75, 1e-9, n_topic)
    assert ei.value.code == 6
    st.close()
    g1.close()
    g2 = engine.Graph(ctx1, n, ptr, dst, rank=0, world=2)              # a shard, but the context has no communicator
    st = engine.PageRankState(g2, 0.75, 1e-9, n_topic)
    with pytest.raises(SpaghettiError) as ei:                          # nothing to exchange before begin
        st.exchange()
    assert ei.value.code == 6
    st.begin()
    with pytest.raises(SpaghettiError) as ei:
        st.exchange()
    assert ei.value.code == 6 and "communicator" in str(ei.value)
    ctx1.comm_init(engine.Context.comm_unique_id(), 0, 1)              # a communicator of the wrong shape
    with pytest.raises(SpaghettiError) as ei:
        st.exchange()
    assert ei.value.code == 6 and "does not match" in str(ei.value)
    st.close()
    g2.close()


_TIMEOUT_SCRIPT = r"""
import os, sys, time
sys.path.insert(0, {root!r})
from spaghettisearch_amd import SpaghettiError, engine, synth
ctx = engine.Context(0)
ctx.set_option("comm.timeout_ms", 1500)
# 1. a rank that waits for a peer which never comes: SS_ERR_COMM after the bound, not a hang
t0 = time.time()
try:
    ctx.comm_init(engine.Context.comm_unique_id(), 0, 2)
    print("FAIL: init of rank 0 of 2 returned without rank 1"); os._exit(1)
except SpaghettiError as e:
    dt = time.time() - t0
    assert e.code == 8 and "did not join" in str(e), (e.code, str(e))
    assert 1.0 < dt < 30.0, dt
assert ctx.comm_info() == (-1, 0)
# 2. the context is still usable: a world of one joins at once ...
ctx.comm_init(engine.Context.comm_unique_id(), 0, 1)
assert ctx.comm_info() == (0, 1)
# 3. ... and a wait for a stream that does not drain in time (here: a long queue of sweeps, standing in for a collective that a
#    rank never joined) ends with SS_ERR_COMM as well
n, e = 2_000_000, 10_000_000
ptr, dst = synth.rmat_graph_torch(n, e, seed=3)
g = engine.Graph(ctx, n, ptr, dst)
st = engine.PageRankState(g, 0.75, -1.0, synth.topic_sizes(n, 16), max_iter=1 << 30)
st.begin()
ctx.set_option("comm.timeout_ms", 20)
st.step(4000)                      # enqueue only: seconds of work
t0 = time.time()
try:
    ctx.synchronize()
    print("FAIL: synchronize returned"); os._exit(1)
except SpaghettiError as e:
    assert e.code == 8 and "did not drain" in str(e), (e.code, str(e))
    assert time.time() - t0 < 5.0
print("TIMEOUT_PATHS_OK")
sys.stdout.flush()
os._exit(0)                         # the helper thread of step 1 is still waiting for rank 1, the device still busy: leave at once
"""


def test_rendezvous_and_collective_waits_are_bounded():
    """VERDICT r3 #6a: ss_comm_init and the library's waits behind a collective end with SS_ERR_COMM after "comm.timeout_ms".
    In a process of its own: the abandoned rendezvous thread and the busy device must not outlive into the test session."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _TIMEOUT_SCRIPT.format(root=root)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "TIMEOUT_PATHS_OK" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
