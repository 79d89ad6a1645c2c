"""Synthetic ScanNet-shaped scenes, utterances and detected boxes.

Replaces the reference's CPU data pipeline (src/joint_det_dataset.py,
src/visual_data_handlers.py) for tests and benchmarks: there are no datasets or
checkpoints offline.  Everything is seeded by the scene index, so every rank /
test regenerates identical inputs.
"""
import numpy as np

# mean colour the reference subtracts (src/joint_det_dataset.py:83)
_MEAN_RGB = np.array([109.8, 97.2, 83.8], dtype=np.float32) / 256.0


def scene(seed, n_points=50000, variant="room", with_color=True):
    """One (n_points, 6) float32 scene: xyz + centred rgb.

    variant "room": 35 % floor, 35 % walls, 5 % ceiling, 25 % box surfaces, 5 mm
    jitter, random point order; the room is centred on the xy-origin with the
    floor at z = 0, so a few points fall inside FPS's origin-skip ball, as in
    axis-aligned ScanNet scans.  variant "uniform": xyz uniform in the room
    volume (sparse worst case for the ball-query early exit).
    """
    rng = np.random.default_rng(seed)
    W, D, H = rng.uniform(4, 9), rng.uniform(3, 7), rng.uniform(2.4, 3.0)
    n = int(n_points)
    if variant == "uniform":
        xyz = rng.uniform([-W / 2, -D / 2, 0], [W / 2, D / 2, H], size=(n, 3))
    else:
        n_floor, n_wall, n_ceil = int(0.35 * n), int(0.35 * n), int(0.05 * n)
        n_box = n - n_floor - n_wall - n_ceil
        parts = []
        parts.append(np.stack([rng.uniform(-W / 2, W / 2, n_floor),
                               rng.uniform(-D / 2, D / 2, n_floor),
                               np.zeros(n_floor)], 1))
        wall = rng.integers(0, 4, n_wall)
        u = rng.uniform(0, 1, n_wall)
        zw = rng.uniform(0, H, n_wall)
        xw = np.where(wall == 0, -W / 2, np.where(wall == 1, W / 2, (u - 0.5) * W))
        yw = np.where(wall == 2, -D / 2, np.where(wall == 3, D / 2, (u - 0.5) * D))
        parts.append(np.stack([xw, yw, zw], 1))
        parts.append(np.stack([rng.uniform(-W / 2, W / 2, n_ceil),
                               rng.uniform(-D / 2, D / 2, n_ceil),
                               np.full(n_ceil, H)], 1))
        K = int(rng.integers(10, 31))
        centres = np.stack([rng.uniform(-W / 2 + 0.5, W / 2 - 0.5, K),
                            rng.uniform(-D / 2 + 0.5, D / 2 - 0.5, K),
                            np.zeros(K)], 1)
        edges = rng.uniform(0.3, 1.5, (K, 3))
        centres[:, 2] = edges[:, 2] / 2
        which = rng.integers(0, K, n_box)
        face = rng.integers(0, 6, n_box)
        uvw = rng.uniform(-0.5, 0.5, (n_box, 3))
        ax = face // 2
        sign = (face % 2) * 1.0 - 0.5
        uvw[np.arange(n_box), ax] = sign
        parts.append(centres[which] + uvw * edges[which])
        xyz = np.concatenate(parts, 0)
        xyz += rng.normal(0, 0.005, xyz.shape)
    xyz = xyz[rng.permutation(n)]
    out = np.zeros((n, 6 if with_color else 3), dtype=np.float32)
    out[:, :3] = xyz.astype(np.float32)
    if with_color:
        out[:, 3:] = rng.uniform(0, 1, (n, 3)).astype(np.float32) - _MEAN_RGB
    return out


def batch(seeds, n_points=50000, variant="room", with_color=True):
    return np.stack([scene(s, n_points, variant, with_color) for s in seeds], 0)


def utterance_tokens(seed, batch_size, max_len=80, vocab=50265):
    """Token ids + attention mask shaped like RobertaTokenizerFast output
    (<s>=0, </s>=2, pad=1); at least one sample reaches max_len."""
    rng = np.random.default_rng(10_000 + seed)
    lens = rng.integers(12, max_len + 1, batch_size)
    lens[rng.integers(0, batch_size)] = max_len
    ids = np.full((batch_size, max_len), 1, dtype=np.int64)
    mask = np.zeros((batch_size, max_len), dtype=np.int64)
    for i, L in enumerate(lens):
        ids[i, 0] = 0
        ids[i, 1:L - 1] = rng.integers(3, vocab, L - 2)
        ids[i, L - 1] = 2
        mask[i, :L] = 1
    return ids, mask


def detected_boxes(seed, batch_size, max_boxes=132, n_classes=485):
    """det_boxes (B,132,6), det_bbox_label_mask (B,132) bool, det_class_ids (B,132) i64
    (train_dist_mod.py:118-125); first n ~ U{10..60} boxes valid."""
    rng = np.random.default_rng(20_000 + seed)
    boxes = np.zeros((batch_size, max_boxes, 6), dtype=np.float32)
    mask = np.zeros((batch_size, max_boxes), dtype=bool)
    cls = np.zeros((batch_size, max_boxes), dtype=np.int64)
    for i in range(batch_size):
        k = int(rng.integers(10, 61))
        boxes[i, :k, :3] = rng.uniform([-4, -3, 0], [4, 3, 2.5], (k, 3))
        boxes[i, :k, 3:] = rng.uniform(0.2, 2.0, (k, 3))
        mask[i, :k] = True
        cls[i, :k] = rng.integers(0, n_classes, k)
    return boxes, mask, cls


def grounding_targets(seed, batch_size, points_xyz, attention_mask, max_boxes=132, n_tok_classes=256):
    """Ground truth in the layout Joint3DDataset hands to the loss (src/joint_det_dataset.py ->
    models/losses.py:650-680): 1-3 real boxes per scene in the first slots of 132, the five token
    maps (rows sum to 1 over the tokens they name), and the per-point instance labels (-1 =
    background) derived from which target box contains the point.
    points_xyz (B,N,3) float array, attention_mask (B,L) 0/1 array (1 = real token)."""
    rng = np.random.default_rng(30_000 + seed)
    B, N = points_xyz.shape[:2]
    out = {
        "box_label_mask": np.zeros((B, max_boxes), np.float32),
        "center_label": np.zeros((B, max_boxes, 3), np.float32),
        "size_gts": np.ones((B, max_boxes, 3), np.float32),
        "sem_cls_label": np.zeros((B, max_boxes), np.int64),
        "point_instance_label": np.full((B, N), -1, np.int64),
    }
    maps = ["positive_map", "modify_positive_map", "pron_positive_map", "other_entity_map", "rel_positive_map"]
    for k in maps:
        out[k] = np.zeros((B, max_boxes, n_tok_classes), np.float32)
    for b in range(B):
        n = int(rng.integers(1, 4))
        ntok = int(attention_mask[b].sum())
        out["box_label_mask"][b, :n] = 1
        lo, hi = points_xyz[b].min(0), points_xyz[b].max(0)
        for t in range(n):
            c = rng.uniform(lo + 0.2 * (hi - lo), hi - 0.2 * (hi - lo))
            s = rng.uniform(0.4, 1.6, 3)
            out["center_label"][b, t] = c
            out["size_gts"][b, t] = s
            out["sem_cls_label"][b, t] = rng.integers(0, 18)
            inside = np.all(np.abs(points_xyz[b] - c) <= 0.5 * s, axis=1)
            out["point_instance_label"][b, inside] = t
            for k, p_on in zip(maps, (1.0, 0.6, 0.3, 0.3, 0.5)):
                if rng.uniform() < p_on:
                    toks = rng.choice(np.arange(1, max(2, ntok - 1)), size=int(rng.integers(1, 4)), replace=False)
                    out[k][b, t, toks] = 1.0 / len(toks)
    return out
