"""Regenerate BASELINE.md section 5 from the round's evidence in profiles/:  python tools/fill_baseline_results.py r04"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"


def line(name):
    p = os.path.join(ROOT, "profiles", f"{tag}_bench_{name}.json")
    if not os.path.exists(p):
        return None
    ls = [l for l in open(p).read().splitlines() if l.startswith("{")]
    return json.loads(ls[-1]) if ls else None


d = line("default")
rows = []
for name, what in [("default", "headline: fp32, B = 8 x 50 000 points, 256 queries, 80 tokens, full train step, pipelined launch structure"),
                   ("one_graph", "same step, one graph / one stream (sampling and text encoder of the batch being trained inside the step)"),
                   ("fps_in_step", "pipelined, but SA1's sampling inside the step (`--fps-prefetch 0`)"),
                   ("130_tokens", "130-token utterances (configs[4] shape, fp32)"),
                   ("hungarian_loss", "the REAL loss (Hungarian matching + `SetCriterion` + seed objectness) inside the graph (`--loss hungarian`; ~32 launches, `csrc/loss.hip`)"),
                   ("hungarian_loss_torch_form", "the same loss in its element-wise torch form (`EDA_FUSED_LOSS=0`: 690 launches; rounds 1-5)"),
                   ("force_dist_hungarian", "the N > 1 step structure + the real loss (`--force-dist --loss hungarian`): the loss's global box count is formed when the batch arrives (`losses.global_box_count`), the captured step holds no collective"),
                   ("force_dist_sync_bn_native_hungarian", "the reference's whole training configuration on one rank: SyncBatchNorm (peer memory) + DDP step structure + real loss (`--force-dist --sync-bn native --loss hungarian`)"),
                   ("text_encoder_eval", "the frozen text encoder in eval mode (`--text-encoder-mode eval`: no dropout inside it; what rounds 1-5 timed -- the reference trains with it in train mode)"),
                   ("attn_bf16", "configs[2]: bf16-MFMA attention contractions (`--attn-dtype bf16`; separate line, never the headline)"),
                   ("attn_f16_130_tokens", "configs[4] on one GPU: fp16 attention + 130 tokens"),
                   ("split_graphs", "the N > 1 graph structure on one GPU (`--split-graphs`)"),
                   ("force_dist", "the N > 1 step through a one-rank RCCL group (`--force-dist`: collective launched between the graphs)"),
                   ("fps_bucket", "single-workgroup bucket sampler instead of the cluster kernels (`EDA_FPS_BUCKET=1`)"),
                   ("wgrad_fp32_mfma", "grouped weight gradients on fp32 MFMA (`EDA_WGRAD_BF16X3=0`)"),
                   ("heads_per_head", "prediction heads' backward per head (`EDA_BATCHED_HEADS=0`)"),
                   ("stock_roberta", "stock Hugging Face text-encoder forward (`EDA_FAST_ROBERTA=0`)"),
                   ("force_dist_sync_bn_native", "`--force-dist --sync-bn native`: SyncBatchNorm statistics exchanged inside the kernels through peer-mapped memory, the step stays one captured graph"),
                   ("force_dist_sync_bn_collective", "`--force-dist --sync-bn collective`: one RCCL micro-collective per BatchNorm layer and direction (eager launches)"),
                   ("deterministic", "`--deterministic 1`: ordered scatter sums, bit-identical runs"),
                   ("splitk_off", "split contraction of the row products off (`EDA_GEMM_SPLITK=0`)"),
                   ("kc96_off", "96-wide chunks off (`EDA_GEMM_KC96=0`)"),
                   ("mha3_ksplit_off", "key-split and bf16 x 3 attention forward off (`EDA_MHA3=0 EDA_MHA2_KSPLIT=0`)"),
                   ("qproj_off", "`EDA_MHA_QPROJ=0` in the evidence run itself, whose build still had `auto` as its default: the configuration that is HEAD's default (the headline line above is HEAD, timed after the switch on another box)"),
                   ("qproj_auto", "the q-projection inside the attention launch of the short-key sites (`EDA_MHA_QPROJ=auto`: rounds 4-6's default, off since the end of round 6)"),
                   ("frozen_b3_off", "the frozen text encoder's wide layers on fp32-MFMA products instead of pre-split bf16 x 3 planes (`EDA_FROZEN_B3=0`)"),
                   ("mha_bwd_r05_form", "attention backward as in round 5: every split range summed by a second launch, single-buffered short-key variants (`EDA_MHA2_BWD_MERGE=0 EDA_MHA2_BWD_DBUF=0`)"),
                   ("mha4_on", "text -> point cross-attention forward on the keys-per-wave kernel (`EDA_MHA4=1`; not the default)"),
                   ("main_stream_alone_TIMING_ONLY", "TIMING ONLY (stale prefetch results): the main stream with nothing underneath it (`EDA_TIMING_SKIP_SIDE=fps,text`)")]:
    x = line(name)
    if x:
        rows.append(f"| {what} | {x['value']:.1f} | {x['ms_per_step']:.2f} |")
r, rh, rm = d["roofline"], d["roofline_hbm"], d["roofline_mfma"]
cb = d.get("cpu_baseline") or {}
ins = d.get("in_step") or {}
txt = f"""## 5. Results (round {int(tag[1:])}, one MI355X; `profiles/{tag}_*`)

No multi-GPU node was available to any round: the 2 / 4 / 8-GPU points of the metric are UNMEASURED (the N > 1 step is
exercised by `tests/test_two_rank_gpu.py`, `--force-dist` and `--split-graphs` on the one GPU there is).

| configuration (`python bench.py ...`, fresh box, all lines of one run) | scenes/s | ms / 8-scene step |
|---|---|---|
""" + "\n".join(rows) + f"""

Headline line: `{d['metric']}` = **{d['value']:.1f} scenes/s** ({d['ms_per_step']:.2f} ms per step, dtype {d['dtype']}, {d['data']}); the
un-pipelined structure timed in the same run: {ins.get('value', float('nan')):.1f} scenes/s ({ins.get('ms_per_step', float('nan')):.2f} ms).
CPU baseline, same run (kind "{cb.get('kind')}": the reference has no CPU op path): {cb.get('value')} scenes/s on {cb.get('cores')} cores
({cb.get('cpu')}; {cb.get('sample')}).

Roofline objects of the line:
* `roofline` (fixed rule since round 5: the roof-priced single-shape launch with the most MAIN-stream time per step): `{r['kernel']}`
  {r['achieved']} TFLOP/s = {r['frac']} of the fp32 MFMA peak, timed as {r.get('timing')}; PMC traffic {r.get('traffic')} bytes per launch
  against {r.get('alg_flops_per_launch')} flops; {r.get('calls_per_step')} launches = {r.get('ms_per_step')} ms of the step, its family {r.get('family_ms_per_step')} ms.
  rocprofv3's per-shape durations, own kernel next to hipBLASLt: `profiles/{tag}_gemm_shapes.txt`.
* `roofline_hbm`: `{rh['kernel']}` {rh['achieved']} GB/s = {rh['frac']} of 8 TB/s ({rh.get('frac_of_achievable')} of the device-copy kernel's
  {rh.get('achievable_copy_gbs')} GB/s) on the bytes of the training formulation (SURVEY §8d's fused bytes + the pre-activations kept for the
  backward); on SURVEY §8d's algorithmic bytes alone: {rh.get('frac_algorithmic')}.
* `roofline_mfma`: `{rm['kernel']}` {rm['achieved']} TFLOP/s = {rm['frac']} of the fp32 MFMA peak.
* attention per shape (fp32 and bf16): `profiles/{tag}_mha_f32.txt`, `profiles/{tag}_mha_bf16.txt`; the grouped weight gradients
  (bf16 x 3, fp32-accurate): `profiles/{tag}_wgrad_grouped.txt`.
"""
p = os.path.join(ROOT, "BASELINE.md")
s = open(p).read()
i = s.index("## 5. Results")
s = s[:i] + txt
open(p, "w").write(s)
print(txt)
