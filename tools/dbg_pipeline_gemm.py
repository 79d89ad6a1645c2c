"""Reproducer / regression screen for a race that only shows when two queues share the CUs.

The frozen text encoder's graph (side stream of eda_amd/pipeline.py) is replayed underneath the main graphs of the
pipelined step, then once more on an idle GPU: the two hidden-state tensors must be bit-identical.  Round 5 found them
1e-1 apart in one replay of three -- gemm_dma_kernel (csrc/gemm.hip) let a chunk's last ds_read cross the barrier that
hands its LDS stage back to the LDS-DMA, and on a CU shared with another queue's workgroups the DMA piece landed first.
Forcing the DMA-staged kernel for every eligible product (EDA_GEMM_SPLITK=0 EDA_GEMM_DMA=1) made every replay fail.

    python tools/dbg_pipeline_gemm.py [steps]         # prints the largest difference per step; exit code 1 if any
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench  # noqa: E402
import check_graph_vs_eager as C  # noqa: E402
from eda_amd import pipeline  # noqa: E402
from eda_amd.parallel import FlatParams  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    dev = torch.device("cuda", 0)
    model = C.make(0, dev, num_queries=64, num_decoder_layers=2)
    batches = [bench.make_inputs(s, 2, dev, 20000, 24) for s in (0, 5, 9)]
    seq = [batches[i % 3] for i in range(steps + 1)]
    flat = FlatParams(model)

    def backward(loss):
        with flat.deferred_wgrad():
            loss.backward()
        flat.collect_grads()

    def update():
        flat.clip_grad_norm_(0.1)

    loss_fn = lambda ep, batch: bench.synthetic_loss(ep)  # noqa: E731
    main_stream = torch.cuda.Stream()
    main_stream.wait_stream(torch.cuda.current_stream())
    worst = 0.0
    with torch.cuda.stream(main_stream):
        for i in range(2):
            loss = loss_fn(model(seq[i]), seq[i]); backward(loss); update()
        torch.cuda.synchronize()
        pipe = pipeline.PipelinedTrainStep(model, seq[0], loss_fn, backward, update, stream=main_stream,
                                           prefetch="sa1", text_prefetch=True)
        for i in range(steps):
            pipe.step(next_batch=seq[i + 1])
            torch.cuda.synchronize()
            busy = pipe.text_next.clone()
            with torch.cuda.stream(pipe.side):
                pipe.g_text.replay()                     # same tokens, idle GPU
            torch.cuda.synchronize()
            d = float((busy - pipe.text_next).abs().max())
            worst = max(worst, d)
            print("step %d: text encoder under the step vs alone: max |diff| = %.3e" % (i, d))
    print("worst", worst)
    return 1 if worst != 0.0 else 0


if __name__ == "__main__":
    sys.exit(main())
