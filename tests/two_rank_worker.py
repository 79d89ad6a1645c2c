"""One rank of the data-parallel training step on the HIP path (tests/test_two_rank_gpu.py launches it).

    python tests/two_rank_worker.py --world W --rank R --port P --out file.pt [--overlap 1]

All ranks share GPU 0 (the GPU box has one device; RCCL refuses two ranks on one device, so the collectives go through
`gloo` on DEVICE tensors) -- everything else is the product's N > 1 code: `sync_bn.enable()` (the library hook inside
`eda_sa_fused_fwd/bwd_f32` calls `dist.all_reduce` on the workspace; torch-op BatchNorm sites exchange packed sums),
`reserve_cus_for_collectives`, `FlatParams.deferred_wgrad()` + flat all-reduce (`--overlap 1`: the two-range
all-reduce started underneath the grouped weight-gradient kernel), global-norm clip, fused AdamW.  world = 1 is the
reference run on the whole global batch.  Scenes of the global batch are sharded contiguously
(main_utils.py:229-240 DistributedSampler-equivalent); Dropout is off so that W ranks x S scenes and 1 rank x W*S
scenes compute the same function (masks are drawn per rank and row)."""
import argparse
import os
import sys

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.dirname(os.path.abspath(__file__))]

import torch
import torch.distributed as dist


def _peer_kind():
    from eda_amd import _lib
    return int(_lib.lib().eda_peer_alloc_kind())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, required=True)
    ap.add_argument("--rank", type=int, required=True)
    ap.add_argument("--port", type=int, default=29611)
    ap.add_argument("--global-batch", type=int, default=8)
    ap.add_argument("--points", type=int, default=6000)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--overlap", type=int, default=0)
    ap.add_argument("--deterministic", type=int, default=0)
    ap.add_argument("--sync", choices=["collective", "native"], default="collective")
    ap.add_argument("--graph", type=int, default=0, help="1: steps 2.. replayed from a captured hipGraph (native sync only)")
    ap.add_argument("--selftest-inject-rank", type=int, default=-1,
                    help="this rank publishes a wrong tag in the peer-memory self-test of sync_bn.enable(native=True)")
    ap.add_argument("--out", required=True)
    a = ap.parse_args()

    import bench
    import model_fixtures as MF
    from eda_amd import attention, parallel, sync_bn
    from eda_amd.bdetr import BeaUTyDETR
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    if a.deterministic:
        from eda_amd import deterministic
        deterministic.enable(True)
    if a.world > 1:
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{a.port}", rank=a.rank, world_size=a.world)
        sync_bn._selftest_inject = (a.rank == a.selftest_inject_rank)
        sync_bn.enable(native=(a.sync == "native"))
        assert sync_bn.fused_hook_installed()
        assert sync_bn.native() == (a.sync == "native" and a.selftest_inject_rank < 0)
        parallel.reserve_cus_for_collectives(32)
        parallel.sampler_without_co_residency()

    torch.manual_seed(0)
    model = BeaUTyDETR(num_queries=64, num_decoder_layers=2, butd=True)
    model.text_encoder = MF.small_roberta(1)
    for p in model.text_encoder.parameters():
        p.requires_grad = False
    MF.fill_det_state(model, seed=41)
    model.to(dev).train()
    model.text_encoder.eval()
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if isinstance(getattr(m, "dropout", None), float):
            m.dropout = 0.0
    flat = parallel.FlatParams(model, parallel.reference_lr_groups)
    # (small steps: the fixture weights are not a trained state, and AdamW's first updates are sign-like -- at the
    # reference's learning rates three steps on them are chaotic and two correct runs drift apart by per cent)
    lrs = {"base": 1e-5, "backbone_net": 1e-5, "text_encoder": 1e-5}
    param0 = flat.flat_param.detach().cpu().clone()
    opt = torch.optim.AdamW([{"params": [gp], "lr": lrs[k]} for k, gp in flat.groups.items()], weight_decay=5e-4, fused=True)

    G = a.global_batch
    S = G // a.world
    full = bench.make_inputs(0, G, dev, a.points, 24)
    lo, hi = a.rank * S, (a.rank + 1) * S

    def shard(v):
        return {k: shard(x) for k, x in v.items()} if isinstance(v, dict) else v[lo:hi].contiguous()
    inputs = shard(full)

    hooks = [0]
    if a.world > 1:
        inner = dist.all_reduce

        def counting(view):
            hooks[0] += 1
            inner(view)
        sync_bn._reduce = counting          # (the hook's seam: same collective, counted)

    losses, grad0 = [], None

    def fwd_bwd():
        attention.advance_dropout_state(dev)
        loss = bench.synthetic_loss(model(inputs))
        with flat.deferred_wgrad():
            loss.backward()
        flat.collect_grads()
        return loss

    graph, static_loss = None, None
    side = torch.cuda.Stream()
    for step in range(a.steps):
        if a.graph and step >= 1:
            # steps 2.. from a captured hipGraph: the statistics exchange lives INSIDE the kernels (sync native), so the
            # forward + backward of a SyncBatchNorm model captures and replays; the flat all-reduce stays between graphs
            if graph is None:
                torch.cuda.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local"):
                    static_loss = fwd_bwd()
                if a.world > 1:
                    dist.barrier()           # (capture is seconds of host work: start replaying together)
            graph.replay()
            loss = static_loss
            flat.all_reduce_mean(a.world)
        elif a.overlap:
            attention.advance_dropout_state(dev)
            loss = bench.synthetic_loss(model(inputs))
            handle = flat.backward_overlapped(loss, a.world)
            handle.wait()
        else:
            with torch.cuda.stream(side) if a.graph else torch.cuda.stream(torch.cuda.current_stream()):
                loss = fwd_bwd()
            if a.graph:
                torch.cuda.current_stream().wait_stream(side)
            flat.all_reduce_mean(a.world)
        if step == 0:
            grad0 = flat.flat_grad.detach().cpu().clone()
            # (running statistics after ONE forward: later steps see parameters that two correct runs have moved apart)
            bn = {k: v.detach().cpu().clone() for k, v in model.state_dict().items() if "running_" in k and "backbone_net.sa1" in k}
        flat.clip_grad_norm_(0.1)
        opt.step()
        losses.append(loss.detach().cpu().clone())
    torch.cuda.synchronize()
    torch.save({"losses": torch.stack(losses), "grad0": grad0, "param": flat.flat_param.detach().cpu(), "bn": bn, "param0": param0,
                "fused_hook_calls": hooks[0], "peer_timeouts": sync_bn.peer_timeouts() if a.sync == "native" and a.world > 1 else 0,
                "captured": graph is not None, "native": sync_bn.native(),
                "peer_alloc_kind": _peer_kind() if a.sync == "native" and a.world > 1 else -1}, a.out)
    if a.world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
