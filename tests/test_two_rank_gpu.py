"""Two data-parallel ranks on the HIP path: 2 processes x 4 scenes on GPU 0 against 1 process x
8 scenes -- `sync_bn.enable()` (the library hook calls `dist.all_reduce` from inside `eda_sa_fused_fwd/bwd_f32`),
`reserve_cus_for_collectives`, deferred weight gradients, flat all-reduce (plain and overlapped with the second half of
the weight-gradient kernel), global-norm clip, fused AdamW.  The GPU box has ONE device and RCCL refuses two ranks on
one device, so the collectives run through `gloo` on device tensors; the product code around them is what runs at
N > 1 (tests/two_rank_worker.py)."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    return port


def _run(world, out_dir, overlap, tag, deterministic=0, sync="collective", graph=0, inject_rank=-1, spin_log2="22"):
    port = _free_port()
    procs, outs = [], []
    env = dict(os.environ, EDA_PEER_SPIN_LOG2=spin_log2)      # (a dead peer costs the test ~5 s, not the production default)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    for r in range(world):
        out = os.path.join(out_dir, f"{tag}_{r}.pt")
        outs.append(out)
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "two_rank_worker.py"), "--world", str(world),
                                       "--rank", str(r), "--port", str(port), "--overlap", str(overlap), "--out", out,
                                       "--deterministic", str(deterministic), "--sync", sync, "--graph", str(graph),
                                       "--selftest-inject-rank", str(inject_rank)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = []
    for p in procs:
        try:
            logs.append(p.communicate(timeout=600)[0])
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    for p, lg in zip(procs, logs):
        assert p.returncode == 0, lg[-3000:]
    return [torch.load(o) for o in outs]


@pytest.mark.parametrize("overlap", [0, 1])
def test_two_ranks_on_the_hip_path_equal_one_rank_on_the_global_batch(tmp_path, overlap):
    one, = _run(1, str(tmp_path), 0, "w1")
    two = _run(2, str(tmp_path), overlap, f"w2o{overlap}")
    # the fused SA / FP calls really exchanged their statistics through the library hook
    assert all(t["fused_hook_calls"] > 0 for t in two) and one["fused_hook_calls"] == 0
    # replicas: identical reduced gradient and identical parameters after three optimizer steps
    assert torch.equal(two[0]["grad0"], two[1]["grad0"])
    assert torch.equal(two[0]["param"], two[1]["param"])
    # loss: the global-batch loss is the mean of the ranks' losses (every term is a mean over scenes)
    l2 = (two[0]["losses"] + two[1]["losses"]) / 2
    g1, g2 = one["grad0"].double(), two[0]["grad0"].double()
    p0, p1, p2 = one["param0"].double(), one["param"].double(), two[0]["param"].double()
    report = dict(losses_two=l2.tolist(), losses_one=one["losses"].tolist(),
                  grad_rel=float((g1 - g2).norm() / g1.norm()), grad_max_rel=float((g1 - g2).abs().max() / g1.abs().max()),
                  update_rel=float((p1 - p2).norm() / (p1 - p0).norm()))
    print(report)
    assert torch.equal(one["param0"], two[0]["param0"])
    assert torch.allclose(l2[:1], one["losses"][:1], rtol=1e-5), report
    assert torch.allclose(l2[:2], one["losses"][:2], rtol=2e-3), report
    # (later steps run on parameters that already differ by summation-order noise, and a step's loss is a discontinuous
    #  function of them -- query top-k, pooling arg-max: a decision that falls the other way moves a loss by up to ~1 %,
    #  as tests/test_graph_gpu.py and tests/test_pipeline_gpu.py record; a wrong reduction shows in step 1 and in the
    #  gradient / update bounds below)
    assert torch.allclose(l2, one["losses"], rtol=2e-2), report
    # reduced flat gradient of the first step = the single-rank gradient (fp32 noise; a handful of ReLU / max-pool
    # decisions may flip between two evaluations, hence the norm-relative bound)
    # (measured: 3.1e-3 of the norm, 0.9 % of the largest entry -- the per-query top-k selection and the ReLU / max-pool
    # decisions of two evaluations differ in a few places; a wrong 1/world or a missing range would give O(1))
    assert report["grad_rel"] <= 1e-2 and report["grad_max_rel"] <= 3e-2, report
    # SyncBN running statistics of SA1 (global-batch mean / unbiased variance)
    for k, v in one["bn"].items():
        assert torch.allclose(two[0]["bn"][k], v, rtol=1e-4, atol=1e-6), k
    # the three optimizer steps moved the parameters the same way (AdamW's first steps are sign-like, so entries whose
    # gradient is rounding noise around zero move by +-lr in either run: the bound is on the update, not per element; a
    # wrong 1/world, a missing range or an un-reduced half would give O(1) here)
    assert report["update_rel"] <= 0.2, report


def test_two_ranks_equal_one_rank_to_rounding_in_the_deterministic_mode(tmp_path):
    """with the ordered scatter sums (eda_amd.deterministic) the only differences between 2 x 4 scenes
    and 1 x 8 scenes are the association of the global sums (BatchNorm statistics rank by rank, weight gradients per rank
    then all-reduced): the bounds that the atomics noise forced open close by two orders of magnitude -- a 5 % scaling
    error in one range of the all-reduce, or a wrong weight decay, no longer fits."""
    one, = _run(1, str(tmp_path), 0, "d1", deterministic=1)
    two = _run(2, str(tmp_path), 1, "d2", deterministic=1)
    assert torch.equal(two[0]["grad0"], two[1]["grad0"]) and torch.equal(two[0]["param"], two[1]["param"])
    l2 = (two[0]["losses"] + two[1]["losses"]) / 2
    g1, g2 = one["grad0"].double(), two[0]["grad0"].double()
    p0, p1, p2 = one["param0"].double(), one["param"].double(), two[0]["param"].double()
    report = dict(losses_two=l2.tolist(), losses_one=one["losses"].tolist(),
                  grad_rel=float((g1 - g2).norm() / g1.norm()), grad_max_rel=float((g1 - g2).abs().max() / g1.abs().max()),
                  update_rel=float((p1 - p2).norm() / (p1 - p0).norm()))
    print("REPORT", report)
    # every loss of the three steps to 1e-4 (the default mode needs 2e-2 on the later ones)
    assert torch.allclose(l2, one["losses"], rtol=1e-4), report
    # the reduced gradient: what is left is not summation noise but a handful of ReLU / max-pool / top-k DECISIONS that
    # the two associations of the global BatchNorm sums flip on the fixture's untrained weights (measured 2.5e-3 of the
    # norm, 4.3e-3 of the largest entry -- the same in every run now, where the default mode scatters around 3e-3 / 9e-3)
    assert report["grad_rel"] <= 5e-3 and report["grad_max_rel"] <= 1e-2, report
    # three optimizer steps: 1 % of the update's norm (measured 0.97 %; the default mode's bound is 20 %): a wrong weight
    # decay or a 5 % scaling error in one range of the all-reduce no longer fits
    assert report["update_rel"] <= 0.03, report


@pytest.mark.parametrize("graph", [0, 1])
def test_two_ranks_with_the_in_kernel_statistics_exchange(tmp_path, graph):
    """SyncBatchNorm without a collective.  Two processes on device 0 map each other's slab
    (csrc/peer.hip) and every BatchNorm kernel exchanges its sums there itself (csrc/peer.h): every site stays on its fused
    kernel, and with graph = 1 steps 2 and 3 are REPLAYS of a captured forward + backward -- global-batch statistics inside a
    hipGraph.  Against one rank on the global batch, in the deterministic mode (so that the bounds are the tight ones)."""
    one, = _run(1, str(tmp_path), 0, "n1", deterministic=1)
    two = _run(2, str(tmp_path), 0, f"n2g{graph}", deterministic=1, sync="native", graph=graph)
    assert all(t["peer_timeouts"] == 0 for t in two), [t["peer_timeouts"] for t in two]
    # the slab other GPUs write and poll while kernels run is fine-grained (or uncached) device memory, never plain hipMalloc
    assert all(t["native"] and t["peer_alloc_kind"] in (0, 1) for t in two), [(t["native"], t["peer_alloc_kind"]) for t in two]
    assert all(t["captured"] == bool(graph) for t in two)
    assert all(t["fused_hook_calls"] == 0 for t in two)            # no collective ran for the statistics
    assert torch.equal(two[0]["grad0"], two[1]["grad0"]) and torch.equal(two[0]["param"], two[1]["param"])
    l2 = (two[0]["losses"] + two[1]["losses"]) / 2
    g1, g2 = one["grad0"].double(), two[0]["grad0"].double()
    p0, p1, p2 = one["param0"].double(), one["param"].double(), two[0]["param"].double()
    report = dict(losses_two=l2.tolist(), losses_one=one["losses"].tolist(), grad_rel=float((g1 - g2).norm() / g1.norm()),
                  update_rel=float((p1 - p2).norm() / (p1 - p0).norm()))
    print("REPORT", report)
    assert torch.allclose(l2, one["losses"], rtol=1e-4), report
    assert report["grad_rel"] <= 5e-3 and report["update_rel"] <= 0.03, report
    for k, v in one["bn"].items():
        assert torch.allclose(two[0]["bn"][k], v, rtol=1e-4, atol=1e-6), k


def test_a_failed_peer_self_test_on_one_rank_sends_both_ranks_to_the_collective_hook(tmp_path):
    """`sync_bn.enable(native=True)` trusts the peer slabs only after one exchange of a known vector.
    Rank 1 publishes a WRONG tag in that self-test: rank 0's polls of rank 1's granules (and rank 1's own) run into their
    bound, the timeout word counts them, the ranks agree on the failure, reset their slabs and BOTH fall back to the
    collective hook -- training then equals one rank on the global batch as in the collective test, with statistics
    exchanged through `dist.all_reduce` (hook calls > 0) and no counted timeout left behind."""
    one, = _run(1, str(tmp_path), 0, "f1", deterministic=1)
    two = _run(2, str(tmp_path), 0, "f2", deterministic=1, sync="native", graph=0, inject_rank=1, spin_log2="12")
    assert all(not t["native"] for t in two)
    assert all(t["fused_hook_calls"] > 0 for t in two)
    assert all(t["peer_timeouts"] == 0 for t in two)                 # (reset after the failed self-test)
    assert torch.equal(two[0]["grad0"], two[1]["grad0"]) and torch.equal(two[0]["param"], two[1]["param"])
    l2 = (two[0]["losses"] + two[1]["losses"]) / 2
    assert torch.allclose(l2, one["losses"], rtol=1e-4), (l2, one["losses"])
