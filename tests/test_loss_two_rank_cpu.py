"""The training loss at N > 1 on CPU (two gloo ranks, different scenes per rank): its normaliser is the GLOBAL batch's box count
(reference models/losses.py:630-636: all-reduced inside the loss).  losses.global_box_count() forms the same number from the
targets alone, outside the step, and end_points["num_boxes_global"] hands it over -- the loss then runs no collective (what a
captured HIP graph needs, bench.py --force-dist --loss hungarian): same loss, same gradients."""
import os
import sys

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
    import loss_fixtures as LF
    import test_losses as TL
    losses, crit = TL._criterion()
    out = {}
    for mode in ("in_loss", "given"):
        ep = LF.make_end_points(40 + rank, B=2 + rank, dataset="scanrefer")          # 2 and 3 scenes: unequal shards
        for k in LF.GRAD_KEYS:
            ep[k].requires_grad_(True)
        assign = TL._oracle_assign(losses, crit, ep)
        if mode == "given":
            ep["num_boxes_global"] = losses.global_box_count(ep["box_label_mask"])
            calls = []
            real = dist.all_reduce
            dist.all_reduce = lambda *a, **k: calls.append(1) or real(*a, **k)
        loss, ep = losses.compute_hungarian_loss(ep, 2, crit, query_points_obj_topk=5, assign=assign)
        loss.backward()
        if mode == "given":
            dist.all_reduce = real
            assert not calls, "the loss ran a collective although its normaliser was given"
        out[mode] = (loss.detach(), {k: ep[k].grad.clone() for k in LF.GRAD_KEYS},
                     float(ep.get("num_boxes_global", torch.zeros(1))[0]), float((ep["box_label_mask"] > 0).sum()))
    torch.save(out, os.path.join(out_dir, f"loss_{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_given_global_box_count_equals_the_in_loss_all_reduce(tmp_path):
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r = [torch.load(tmp_path / f"loss_{i}.pt") for i in range(2)]
    total = r[0]["given"][3] + r[1]["given"][3]
    for i in range(2):
        assert r[i]["given"][2] == total and r[i]["given"][3] < total          # the global count, not the rank's own
        assert torch.equal(r[i]["in_loss"][0], r[i]["given"][0])
        for k, g in r[i]["in_loss"][1].items():
            assert torch.equal(g, r[i]["given"][1][k]), k
