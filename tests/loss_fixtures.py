"""Deterministic synthetic inputs for the Hungarian-loss parity tests (shared by the golden
generator, which feeds them to the REFERENCE's models/losses.py in the build container, and by
the tests, which feed them to eda_amd/losses.py)."""
import numpy as np
import torch

PREFIXES = ["proposal_", "last_", "0head_"]        # num_decoder_layers = 2
G = 132                                             # padded ground-truth slots (reference data layout)


def make_end_points(seed, B=3, Q=32, L=20, C=256, K=64, npts=400, dataset="scanrefer"):
    """end_points with everything compute_hungarian_loss reads: predictions for three prefixes,
    padded targets (1-3 valid boxes per scene) with the five token maps, seed-objectness inputs."""
    rng = np.random.default_rng(seed)
    f = lambda *s: torch.from_numpy(rng.standard_normal(s).astype(np.float32))
    u = lambda lo, hi, *s: torch.from_numpy(rng.uniform(lo, hi, s).astype(np.float32))
    ep = {"language_dataset": [dataset] * B}
    for p in PREFIXES:
        ep[f"{p}center"] = u(-2, 2, B, Q, 3)
        ep[f"{p}pred_size"] = u(0.2, 1.5, B, Q, 3)
        ep[f"{p}sem_cls_scores"] = f(B, Q, C)
        pq = f(B, Q, 64)
        ep[f"{p}proj_queries"] = pq / pq.norm(dim=-1, keepdim=True)
    pt = f(B, L, 64)
    ep["proj_tokens"] = pt / pt.norm(dim=-1, keepdim=True)
    lens = rng.integers(L // 2, L + 1, B)
    lens[0] = L
    ep["tokenized"] = {"attention_mask": torch.from_numpy((np.arange(L)[None] < lens[:, None]).astype(np.int64))}
    nt = rng.integers(1, 4, B)                       # valid targets per scene
    mask = torch.zeros(B, G)
    for b in range(B):
        mask[b, :nt[b]] = 1
    ep["box_label_mask"] = mask
    ep["center_label"] = u(-2, 2, B, G, 3)
    ep["size_gts"] = u(0.2, 1.5, B, G, 3)
    ep["sem_cls_label"] = torch.from_numpy(rng.integers(0, 18, (B, G)).astype(np.int64))

    def token_map(p_on):
        m = (rng.uniform(0, 1, (B, G, C)) < p_on).astype(np.float32)
        m[:, :, L:] = 0                              # only real token positions
        m[:, :, C - 1] = 0
        s = m.sum(-1, keepdims=True)
        return torch.from_numpy(np.where(s > 0, m / np.maximum(s, 1), 0).astype(np.float32))
    ep["positive_map"] = token_map(0.15)
    for b in range(B):                               # every valid target names at least one token
        for g in range(nt[b]):
            if ep["positive_map"][b, g].sum() == 0:
                ep["positive_map"][b, g, 1 + g] = 1.0
    ep["modify_positive_map"] = token_map(0.05)
    ep["pron_positive_map"] = token_map(0.03)
    ep["other_entity_map"] = token_map(0.04)
    ep["rel_positive_map"] = token_map(0.04)
    ep["auxi_entity_positive_map"] = token_map(0.05)[:, :1]
    ep["auxi_box"] = torch.cat([u(-2, 2, B, 1, 3), u(0.2, 1.5, B, 1, 3)], -1)
    # seed objectness branch
    ep["seed_inds"] = torch.from_numpy(np.stack([rng.permutation(npts)[:K] for _ in range(B)]).astype(np.int32))
    ep["seed_xyz"] = u(-2, 2, B, K, 3)
    ep["seeds_obj_cls_logits"] = f(B, 1, K)
    pil = rng.integers(-1, 3, (B, npts)).astype(np.int64)
    ep["point_instance_label"] = torch.from_numpy(pil)
    return ep


GRAD_KEYS = [f"{p}{k}" for p in PREFIXES for k in ("center", "pred_size", "sem_cls_scores", "proj_queries")] + [
    "proj_tokens", "seeds_obj_cls_logits"]
