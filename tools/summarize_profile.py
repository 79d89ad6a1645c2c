"""profiles/<tag>_summary.md from a rocprofv3 --kernel-trace --stats CSV of an eager bench run
plus the bench JSON lines:  python tools/summarize_profile.py TAG STEPS kernel_stats.csv default.json rocprof.json"""
import csv
import json
import re
import sys

NATIVE = [
    ("native: own row GEMMs fwd / dX, tiled (gemm_rows / gemm_dma: linear layers incl. the text encoder's and the LayerNorm-epilogue launches, small SA/FP layers)", r"^gemm_rows_kernel|^gemm_dma_kernel|^gemm_b3_rows_kernel"),
    ("native: the frozen text encoder's wide linear layers on pre-split bf16 x 3 weight planes (gemm_frozen.hip)", r"^linear_frozen_b3_kernel|^bf16x3_split"),
    ("native: own row GEMMs fwd / dX, streaming (gemm_stream, fp32 MFMA or bf16 x 3 / gemm_gather3: many-row SA layers)", r"^gemm_stream_kernel|^gemm_stream_b3_kernel|^gemm_gather3"),
    ("native: furthest point sampling", r"^fps_"),
    ("native: fused attention (fwd, dQ, dK/dV)", r"^mha[234]?_"),
    ("native: BN+ReLU(+pool) fwd/bwd", r"^bn_"),
    ("native: residual+dropout+LayerNorm", r"^add_dropout_ln|^ln_reduce"),
    ("native: weight/bias gradients (grouped wgrad fp32 / bf16 x 3, wgrad_x, colsum, the heads' 3-channel layers) and the SA layers' one-launch backward (sa_layer_bwd: weight + input gradient)", r"^wgrad_|^colsum_|^wcolsum_|^weight_transpose|^transpose_batch|^tiny_out_|^sa_layer_bwd|^sa_gather_layer_bwd|^rows_scatter_add"),
    ("native: small fused kernels (ordered / n-ary adds, l2norm of the contrastive projections, device assignment, peer exchange)", r"^add_n_kernel|^l2norm_|^lsa_|^peer_|^det_scatter|^copy_kernel|^sa_eval|^center_query_pos"),
    ("native: ball query (grid build + query)", r"^gq_|^ball_query"),
    ("native: gather/group/3-NN", r"^group_|^gather_|^three_"),
    ("native: clip + AdamW on the flat buffers (optim.hip)", r"^grad_sumsq_kernel|^adamw_flat_kernel"),
    ("native: the training loss (loss.hip; --loss hungarian only) and the frozen text encoder's dropout", r"^compact_targets|^match_|^box_loss|^pos_align|^sem_align|^seed_objectness|^loss_combine|^scale_by_scene|^dropout_flat"),
    ("native: zero-fill", r"^zero_kernel"),
]
TORCH = [
    ("library GEMM (hipBLASLt, fp32: the batched query x token products of the contrastive losses; the text encoder too with EDA_FAST_ROBERTA=0)", r"^Cijk_|gemm|Gemm"),
    ("torch foreach (gradient gather, batch rotation; rounds 1-5: also the optimizer)", r"multi_tensor_apply"),
    ("torch reduce", r"reduce_kernel"),
    ("torch layernorm / softmax / attention (RoBERTa)", r"layer_norm|softmax|attn_fwd|LayerNorm"),
    ("torch copy/cat", r"copy|Copy|CatArray"),
    ("fill/memset", r"fill|Fill|memset"),
    ("torch elementwise / dropout / index", r"elementwise|dropout|index|gather|scatter|embedding"),
]


def short(n):
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"at::native::|\(anonymous namespace\)::", "", n)
    return n.split("(")[0][:70]


def main():
    tag, steps, stats, dflt, under = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
    rows = list(csv.DictReader(open(stats)))
    fam, native_rows = {}, []
    tot_ms = tot_n = 0.0
    for r in rows:
        name = short(r["Name"])
        ms = float(r["TotalDurationNs"]) / 1e6 / steps
        n = int(r["Calls"]) / steps
        tot_ms += ms
        tot_n += n
        for label, pat in NATIVE + TORCH:
            if re.search(pat, name):
                break
        else:
            label = "other"
        f = fam.setdefault(label, [0.0, 0.0])
        f[0] += ms
        f[1] += n
        if label.startswith("native"):
            native_rows.append((name, n, float(r["AverageNs"]) / 1e3, ms))
    d = json.load(open(dflt))
    u = json.load(open(under))
    cb, rf, rm, rh = d.get("cpu_baseline") or {"value": None, "unit": "", "cores": None, "sample": "not run"}, d["roofline"], d.get("roofline_mfma"), d.get("roofline_hbm")
    out = [f"# Run {tag} — B=8 x 50 000 points, 256 queries, 80 tokens, fp32", ""]
    out.append(f"`python bench.py` (launch: {d['config'].get('launch')}; SA1 sampling: {d['config'].get('sa1_sampling', 'inside the step')}): **{d['value']} scenes/s, "
               f"{d['ms_per_step']} ms/step**; cpu_baseline {cb['value']} {cb['unit']} on {cb['cores']} cores "
               f"({cb['sample']}).  Full line: `{tag}_bench_default.json`.")
    out.append("")
    out.append(f"roofline (the roofline-priced native kernel with the most time per step, bound {rf['bound']}): "
               f"`{rf.get('kernel')}` {rf['achieved']} {rf['unit']} = {rf['frac']} of {rf['peak']}"
               + (f"; roofline_hbm: `{rh.get('kernel')}` {rh['achieved']} {rh['unit']} = {rh['frac']} of {rh['peak']}"
                  if rh else "")
               + (f"; roofline_mfma: `{rm.get('kernel')}` {rm['achieved']} {rm['unit']} = "
                  f"{rm['frac']} of {rm['peak']} (fp32 MFMA)." if rm else "."))
    fps = d.get("fps") or []
    if fps:
        out.append("")
        out.append("fps (latency-bound, us per dependent round): "
                   + ", ".join(f"N={f['n']}->m={f['m']}: {f['ms']} ms = {f['us_per_round']} us/round" for f in fps) + ".")
    out.append("")
    out.append(f"`rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps {steps - 3} --warmup 3 "
               f"--graph 0 --cpu-scenes 0` (eager, {steps} steps; under the profiler {u['ms_per_step']} ms/step): "
               f"{tot_ms:.1f} ms of kernels per step over {tot_n:.0f} launches (`{tag}_bench_eager_kernel_stats_rocprofv3.csv`).")
    out += ["", "## Where the step goes",
            "", "(kernel time of ALL streams; with the default launch structure the frozen text encoder -- the library GEMM and "
            "`torch layernorm / softmax / attention` rows plus ~0.4 ms of the element-wise row -- and SA1's sampling "
            "(`fps_spec_kernel`) run on the second stream, off the critical path)",
            "", "| family | ms/step | launches/step |", "|---|---|---|"]
    for label, (ms, n) in sorted(fam.items(), key=lambda kv: -kv[1][0]):
        out.append(f"| {label} | {ms:.2f} | {n:.0f} |")
    out += ["", "## Native kernels (avg duration agrees with bench.py's HIP-event numbers in `kernels`)", "",
            "| kernel | calls/step | avg us | ms/step |", "|---|---|---|---|"]
    for name, n, avg, ms in sorted(native_rows, key=lambda r: -r[3]):
        out.append(f"| `{name}` | {n:.1f} | {avg:.1f} | {ms:.3f} |")
    print("\n".join(out))


if __name__ == "__main__":
    main()
