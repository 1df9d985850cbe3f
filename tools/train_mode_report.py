"""Prints how close the train-mode forward is to the reference fixture (tests/golden/train_mode.npz): max |line_desc - reference| of both
calls and the largest relative running-statistics error.  On the GPU box:  python tools/train_mode_report.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import load, train_mode_batches  # noqa: E402
from workloads import synth  # noqa: E402

torch.set_grad_enabled(False)


def main():
    from models.line_process import line_tokenizer
    from models.line_transformer import LineTransformer
    g = load("train_mode")
    nl = int(g["n_desc_layers"])
    m = LineTransformer({"mode": "train", "max_keylines": -1, "min_length": 16, "token_distance": 8, "nn_threshold": 0.8,
                         "n_line_descriptive_layers": nl})
    m.load_state_dict(synth.to_torch_state_dict(synth.calibrated_state_dict(nl)), strict=True)
    m = m.to("cuda").eval()
    hw = tuple(int(v) for v in g["hw"])
    pre = lambda rows, pred: m.preprocess(synth.array_to_keylines(rows), (1, 1, *hw), pred)
    tok = lambda lines, pred: line_tokenizer(lines, 8, 21, pred, (640, 480))
    batches = train_mode_batches(g, pre, tok, to_dev=lambda t: t.cuda())
    m.train()
    m.dropout = 0.0
    for c, batch in enumerate(batches):
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        got = m(batch)["line_desc"]
        t1.record(); torch.cuda.synchronize()
        got = got.cpu().numpy()
        want = g[f"line_desc_{c}"]
        err = float(np.abs((got if c == 0 else got[:, :, ::5]) - want).max())
        rel = 0.0
        for k, v in m.state_dict().items():
            if "running_" in k:
                ref = g[f"bn{c}.{k}"]
                rel = max(rel, float(np.abs(v.cpu().numpy() - ref).max() / max(1.0, np.abs(ref).max())))
        print(f"call {c}: max|line_desc - reference| = {err:.3e}, running statistics max relative error = {rel:.3e}, "
              f"forward incl. engine build {t0.elapsed_time(t1):.2f} ms")
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0.record()
    for _ in range(10):
        m(batches[0])
    t1.record(); torch.cuda.synchronize()
    print(f"steady train-mode forward of 3 x 250 sub-lines: {t0.elapsed_time(t1) / 10:.3f} ms per call")


if __name__ == "__main__":
    main()
