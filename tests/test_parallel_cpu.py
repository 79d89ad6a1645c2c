"""The N > 1 path on CPU: two gloo processes, scenes sharded across ranks, one
flat-buffer all-reduce per step.  Checks that the reduced gradient equals the
mean of the per-rank gradients and that replicas stay bit-identical after
optimizer steps.  (Native ops come from the oracle façade, injected by the test.)"""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle_ext
    from eda_amd import pointnet2_utils
    pointnet2_utils._ext = oracle_ext          # CPU test doubles for the HIP ops
    from eda_amd import attention
    from oracle import attention_ref
    attention._core = attention_ref.attention_core
    from eda_amd.parallel import FlatParams, broadcast_parameters, shard_scene_seeds
    from eda_amd.pointnet2_modules import PointnetSAModuleVotes
    from eda_amd.encoder_decoder_layers import BiDecoderLayer
    from eda_amd import synthetic

    torch.manual_seed(100 + rank)              # different init per rank on purpose
    sa = PointnetSAModuleVotes(npoint=64, radius=0.4, nsample=8, mlp=[3, 16, 32], use_xyz=True, normalize_xyz=True)
    dec = BiDecoderLayer(32, n_heads=4, dim_feedforward=64, dropout=0.0, self_position_embedding="xyz_learned", butd=False)
    model = torch.nn.ModuleList([sa, dec]).train()
    broadcast_parameters(model, 0)
    grads = FlatParams(model, lambda n: "sa" if n.startswith("0.") else "dec")
    assert set(grads.groups) == {"sa", "dec"}
    opt = torch.optim.AdamW([{"params": [gp], "lr": 1e-2} for gp in grads.groups.values()])

    seeds = shard_scene_seeds(4, rank, world)            # global batch 4 -> 2 scenes per rank
    pc = torch.from_numpy(synthetic.batch(seeds, 1500))
    xyz, feats = pc[..., :3].contiguous(), pc[..., 3:].transpose(1, 2).contiguous()
    lang = torch.randn(2, 5, 32, generator=torch.Generator().manual_seed(rank))
    mask = torch.zeros(2, 5, dtype=torch.bool)

    def local_loss():
        new_xyz, f, _ = sa(xyz, feats)
        q = dec(f.transpose(1, 2).contiguous(), f.transpose(1, 2).contiguous(), lang, new_xyz, None, mask)
        return q.pow(2).mean()

    for it in range(3):
        if it == 2:
            # the overlapped form (FlatParams.backward_overlapped: range-wise flush + gather with the first range's
            # all-reduce in flight underneath the second range): same reduced gradient as the plain path
            with grads.deferred_wgrad():
                local_loss().backward()
            grads.collect_grads()
            local = grads.flat_grad.clone()
            mid = grads.split_offset(0.5)
            assert 0 < mid < local.numel()
            grads.backward_overlapped(local_loss(), world).wait()
            gathered = [torch.zeros_like(local) for _ in range(world)]
            dist.all_gather(gathered, local)
        else:
            local_loss().backward()
            grads.collect_grads()
            assert all(p.grad is None for p in grads.params)
            local = grads.flat_grad.clone()
            gathered = [torch.zeros_like(local) for _ in range(world)]
            dist.all_gather(gathered, local)
            if it == 0:
                grads.all_reduce_mean(world)
            else:                                           # the asynchronous form: a mean only after wait()
                grads.all_reduce_mean(world, async_op=True).wait()
        assert all(p.grad is None for p in grads.params)
        expect = sum(gathered) / world
        assert torch.allclose(grads.flat_grad, expect, rtol=1e-6, atol=1e-7), "all-reduce != mean of rank grads"
        assert not torch.equal(gathered[0], gathered[1]), "ranks saw the same scenes"
        grads.clip_grad_norm_(10.0)
        before = grads.params[0].detach().clone()
        opt.step()
        assert not torch.equal(before, grads.params[0].detach()), "flat update did not reach the module parameter"
    flat_params = torch.cat([p.detach().reshape(-1) for p in grads.params])
    assert torch.equal(flat_params, grads.flat_param)
    torch.save(flat_params, os.path.join(out_dir, f"params_{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_flat_allreduce(tmp_path, oracle):
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    p0 = torch.load(tmp_path / "params_0.pt")
    p1 = torch.load(tmp_path / "params_1.pt")
    assert torch.equal(p0, p1), "replicas diverged"


def test_shard_scene_seeds():
    from eda_amd.parallel import shard_scene_seeds
    allseeds = sum((shard_scene_seeds(64, r, 8) for r in range(8)), [])
    assert allseeds == list(range(64))


# ---- SyncBatchNorm-equivalent statistics (eda_amd/sync_bn.py) ----------------------------------------------
def _sync_bn_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from eda_amd import sync_bn, nn_utils
    sync_bn.enable()
    g = torch.Generator().manual_seed(7)
    C, rows = 16, [300, 212]                       # unequal shards: the count must travel with the sums
    z_all = torch.randn(sum(rows), C, generator=g) * 1.7 + 0.3
    w_all = torch.randn(sum(rows), C, generator=g)
    lo = sum(rows[:rank])
    z = z_all[lo:lo + rows[rank]].clone().requires_grad_(True)
    bn = torch.nn.BatchNorm1d(C)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5, generator=g); bn.bias.normal_(0, 0.2, generator=g)
    bn.train()
    y = nn_utils.bn_relu_rows(bn, z) if False else sync_bn.bn_relu(bn, z)
    (y * w_all[lo:lo + rows[rank]]).sum().backward()
    torch.save({"y": y.detach(), "dz": z.grad, "dgamma": bn.weight.grad, "dbeta": bn.bias.grad,
                "rm": bn.running_mean.clone(), "rv": bn.running_var.clone()}, os.path.join(out_dir, f"sbn{rank}.pt"))
    dist.destroy_process_group()


def test_sync_bn_two_ranks_equal_single_process_bn_over_the_concatenated_batch(tmp_path):
    """Reference semantics at N > 1 (main_utils.py:336-338, SyncBatchNorm): two gloo ranks with unequal
    shards must reproduce ONE BatchNorm over the concatenated rows -- outputs, input gradients, running
    statistics; d(gamma)/d(beta) are local sums whose sum over ranks is the single-process gradient."""
    world, port = 2, 29653
    mp.spawn(_sync_bn_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    g = torch.Generator().manual_seed(7)
    C, rows = 16, [300, 212]
    z_all = (torch.randn(sum(rows), C, generator=g) * 1.7 + 0.3).requires_grad_(True)
    w_all = torch.randn(sum(rows), C, generator=g)
    bn = torch.nn.BatchNorm1d(C)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5, generator=g); bn.bias.normal_(0, 0.2, generator=g)
    bn.train()
    y = torch.relu(bn(z_all))
    (y * w_all).sum().backward()
    outs = [torch.load(os.path.join(str(tmp_path), f"sbn{r}.pt")) for r in range(world)]
    torch.testing.assert_close(torch.cat([o["y"] for o in outs]), y.detach(), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(torch.cat([o["dz"] for o in outs]), z_all.grad, rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(outs[0]["dgamma"] + outs[1]["dgamma"], bn.weight.grad, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(outs[0]["dbeta"] + outs[1]["dbeta"], bn.bias.grad, rtol=1e-4, atol=1e-5)
    for o in outs:
        torch.testing.assert_close(o["rm"], bn.running_mean, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(o["rv"], bn.running_var, rtol=1e-5, atol=1e-6)


def test_dropout_counter_is_seeded_by_torch_and_the_rank(monkeypatch):
    """the dropout counter must follow torch.manual_seed and differ between data-parallel ranks."""
    from eda_amd import attention
    torch.manual_seed(1234)
    monkeypatch.setenv("RANK", "0")
    a0 = attention._initial_dropout_counter()
    monkeypatch.setenv("RANK", "1")
    a1 = attention._initial_dropout_counter()
    torch.manual_seed(1235)
    b1 = attention._initial_dropout_counter()
    torch.manual_seed(1234)
    assert attention._initial_dropout_counter() == a1
    assert len({a0, a1, b1}) == 3 and all(0 <= v < (1 << 62) for v in (a0, a1, b1))
