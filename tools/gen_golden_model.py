"""Generate tests/golden/model_*.npz by RUNNING THE REFERENCE's Python layers
(/root/reference, imported in this build container only -- see tools/ref_import.py)
with deterministic weights (tests/model_fixtures.det_state) and seeded inputs.
Only the resulting arrays are stored.  Fixture sets:
  (2) QueryAndGroup, PointnetSAModuleVotes, PointnetFPModule, Pointnet2Backbone (eval)
  (3) CrossAttentionLayer, BiEncoderLayer, BiEncoder, BiDecoderLayer (eval, butd
      on/off, padded masks): outputs + input gradients of a sum() loss
  (4) one small full-model forward (N=4096, L=16, 64 queries).
Run from the repo root:  python tools/gen_golden_model.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ref_import  # noqa: E402
import model_fixtures as MF  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def save(name, **arrays):
    path = os.path.join(OUT, f"model_{name}.npz")
    np.savez_compressed(path, **MF.pack(arrays))
    print(f"{path}: {os.path.getsize(path) / 1024:.0f} KiB")


def main():
    mods = ref_import.load()
    torch.manual_seed(0)

    # ---- (2) pointnet2 layers ------------------------------------------------
    pc = MF.make_cloud(1, 2, 4096)
    xyz = pc[..., :3].contiguous()
    feats = pc[..., 3:].transpose(1, 2).contiguous()
    inds = mods.pointnet2_utils.furthest_point_sample(xyz, 256)
    new_xyz = mods.pointnet2_utils.gather_operation(xyz.transpose(1, 2).contiguous(), inds).transpose(1, 2).contiguous()
    qg = mods.pointnet2_utils.QueryAndGroup(0.3, 16, use_xyz=True, ret_grouped_xyz=True, normalize_xyz=True)
    nf, gx = qg(xyz, new_xyz, feats)
    save("query_and_group", inds=inds, new_features=nf, grouped_xyz=gx)

    sa = mods.pointnet2_modules.PointnetSAModuleVotes(npoint=256, radius=0.3, nsample=16,
                                                      mlp=[3, 16, 16, 32], use_xyz=True, normalize_xyz=True)
    MF.fill_det_state(sa, seed=2); sa.eval()
    f_in = feats.clone().requires_grad_(True)
    sx, sf, si = sa(xyz, f_in)
    sf.sum().backward()
    save("sa_module", new_xyz=sx, features=sf, inds=si, grad_features=f_in.grad)

    # the same module in TRAIN mode: batch statistics, running-stat update, parameter grads
    sa_t = mods.pointnet2_modules.PointnetSAModuleVotes(npoint=256, radius=0.3, nsample=16,
                                                        mlp=[3, 16, 16, 32], use_xyz=True, normalize_xyz=True)
    MF.fill_det_state(sa_t, seed=2); sa_t.train()
    f_t = feats.clone().requires_grad_(True)
    _, tf, _ = sa_t(xyz, f_t)
    (tf * MF.make_feats(6, *tf.shape)).sum().backward()
    l0, l2 = sa_t.mlp_module.layer0, sa_t.mlp_module.layer2
    save("sa_module_train", features=tf, grad_features=f_t.grad,
         grad_w0=l0.conv.weight.grad, grad_w2=l2.conv.weight.grad,
         grad_gamma0=l0.bn.bn.weight.grad, grad_beta0=l0.bn.bn.bias.grad,
         grad_gamma2=l2.bn.bn.weight.grad, grad_beta2=l2.bn.bn.bias.grad,
         running_mean0=l0.bn.bn.running_mean, running_var0=l0.bn.bn.running_var,
         running_mean2=l2.bn.bn.running_mean, running_var2=l2.bn.bn.running_var,
         nbt=l2.bn.bn.num_batches_tracked)

    fp = mods.pointnet2_modules.PointnetFPModule(mlp=[32 + 8, 24, 16])
    MF.fill_det_state(fp, seed=3); fp.eval()
    unk_f = MF.make_feats(4, 2, 8, 4096).requires_grad_(True)
    kn_f = sf.detach().clone().requires_grad_(True)
    fo = fp(xyz, sx, unk_f, kn_f)
    (fo * MF.make_feats(5, *fo.shape)).sum().backward()
    save("fp_module", out=fo, grad_unknown=unk_f.grad, grad_known=kn_f.grad)

    bb = mods.backbone.Pointnet2Backbone(input_feature_dim=3, width=1)
    MF.fill_det_state(bb, seed=4); bb.eval()
    with torch.no_grad():
        ep = bb(pc, {})
    save("backbone", **{k: v for k, v in ep.items()})

    # ---- (3) encoder / decoder layers -----------------------------------------
    d = 288
    B, V, L, D, Q = 2, 96, 12, 20, 40
    for butd in (True, False):
        tag = "butd" if butd else "nobutd"
        vis = MF.make_feats(10, B, V, d).requires_grad_(True)
        pos = MF.make_feats(11, B, V, d, scale=0.5)
        text = MF.make_feats(12, B, L, d).requires_grad_(True)
        det = MF.make_feats(13, B, D, d).requires_grad_(True) if butd else None
        vmask = torch.zeros(B, V, dtype=torch.bool)
        tmask = MF.make_mask(1, B, L, min_valid=3)
        dmask = MF.make_mask(2, B, D, min_valid=2) if butd else None
        layer = mods.edl.BiEncoderLayer(d, dropout=0.1, activation="relu", n_heads=8, dim_feedforward=256,
                                        self_attend_lang=True, self_attend_vis=True, use_butd_enc_attn=butd)
        enc = mods.edl.BiEncoder(layer, 3)
        MF.fill_det_state(enc, seed=20); enc.eval()
        vo, to = enc(vis, pos, vmask, text, tmask, {}, detected_feats=det, detected_mask=dmask)
        loss = (vo * MF.make_feats(14, *vo.shape)).sum() + (to * MF.make_feats(15, *to.shape)).sum()
        loss.backward()
        arrays = dict(vis_out=vo, text_out=to, grad_vis=vis.grad, grad_text=text.grad,
                      grad_w=enc.layers[1].cross_layer.cross_lv.in_proj_weight.grad)
        if butd:
            arrays["grad_det"] = det.grad
        save(f"biencoder_{tag}", **arrays)

        dec = mods.edl.BiDecoderLayer(d, n_heads=8, dim_feedforward=256, dropout=0.1, activation="relu",
                                      self_position_embedding="loc_learned", butd=butd)
        MF.fill_det_state(dec, seed=21); dec.eval()
        query = MF.make_feats(16, B, Q, d).requires_grad_(True)
        qpos = MF.make_feats(17, B, Q, 6)
        vis2 = MF.make_feats(18, B, V, d).requires_grad_(True)
        lang = MF.make_feats(19, B, L, d).requires_grad_(True)
        det2 = MF.make_feats(20, B, D, d) if butd else None
        qo = dec(query, vis2, lang, qpos, None, tmask, detected_feats=det2, detected_mask=dmask)
        (qo * MF.make_feats(21, *qo.shape)).sum().backward()
        save(f"bidecoder_{tag}", out=qo, grad_query=query.grad, grad_vis=vis2.grad, grad_lang=lang.grad,
             grad_w=dec.cross_v.in_proj_weight.grad)

    full_model(mods)


def full_model(mods, only=None):
    # ---- (4) full model --------------------------------------------------------
    # (tag, butd, padded utterance length): the 130-token case is the SR3D-shaped long utterance of BASELINE.json
    # configs[4] (python tools/gen_golden_model.py --only full_butd_l130 regenerates just that file)
    for tag, butd, max_len in (("butd", True, 16), ("nobutd", False, 16), ("butd_l130", True, 130)):
        if only and only != f"full_{tag}":
            continue
        ref_import.FakeTokenizer.max_len = max_len
        model = ref_import.build_reference_model(mods, seed=0, num_queries=64, butd=butd)
        model.text_encoder = MF.small_roberta(1)
        MF.fill_det_state(model, seed=30)
        model.eval()
        for input_seed in range(3, 20):
            inputs = MF.full_model_inputs(input_seed, max_len=max_len)
            tok = ref_import.FakeTokenized(inputs["tokenized"]["input_ids"],
                                           inputs["tokenized"]["attention_mask"])

            class Tok:
                def batch_encode_plus(self, *a, **k):
                    return tok
            model.tokenizer = Tok()
            ref_inputs = {k: v for k, v in inputs.items() if k != "tokenized"}
            ref_inputs["text"] = ["x"] * 2
            with torch.no_grad():
                model.points_obj_cls.conv3.bias.fill_(-13.0)
                ep = model(ref_inputs)
                # centre the seed-objectness logits (see model_fixtures.det_tensor) and re-run
                bias = round(-13.0 - ep["seeds_obj_cls_logits"].mean().item(), 2)
                model.points_obj_cls.conv3.bias.fill_(bias)
                ep = model(ref_inputs)
            out = {k: v for k, v in ep.items() if torch.is_tensor(v)}
            sc = torch.sigmoid(out["seeds_obj_cls_logits"][:, 0]).sort(dim=1, descending=True)[0][:, :65]
            gap = (sc[:, :-1] - sc[:, 1:]).min().item()
            edge = (sc[:, 63] - sc[:, 64]).min().item()
            print("input seed", input_seed, "min top-k gap", gap, "gap at the 64/65 boundary", edge)
            if gap > 2e-6 and edge > 2e-4:   # a stable query SET (tests align the order)
                break
        else:
            raise SystemExit("no stable fixture seed found")
        out["fixture_obj_cls_bias"] = torch.tensor(bias)
        out["fixture_input_seed"] = torch.tensor(input_seed)
        out["fixture_max_len"] = torch.tensor(max_len)
        save(f"full_{tag}", **out)
        print(len(out), "tensors")


if __name__ == "__main__":
    if len(sys.argv) == 3 and sys.argv[1] == "--only" and sys.argv[2].startswith("full_"):
        full_model(ref_import.load(), only=sys.argv[2])
    else:
        main()
