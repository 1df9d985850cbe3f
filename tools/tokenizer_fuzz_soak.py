"""Extended run of tests/test_gpu_tokenizer_fuzz.py: ~43 000 random detector lines (12 seeds x max_tokens {3, 21, 41} x token_distance
{8, 12.8, 16, 9.7}) through linetr_prefilter_batch + linetr_tokenize, every token tensor (and mat_klines2sublines from the tokeniser launch)
against the CPU oracle.   python tools/tokenizer_fuzz_soak.py   (on the GPU box; r04: 42 883 lines, 0 mismatching tensors)"""
import sys, os, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
from test_gpu_tokenizer_fuzz import fuzz_lines, HW, BORDER, MIN_LEN
from helpers import TOK_KEYS
from oracle import linetr_oracle as O
from workloads import synth
from linetr_amd.engine import Engine
torch.set_grad_enabled(False)
eng = Engine(synth.calibrated_state_dict(), "cuda:0", image_shape=list(HW))
dd, ds = synth.synth_dense_maps(99, *HW)
ddc, dsc = dd.cuda(), ds.cuda()
tot = bad = 0; worst = 0.0
for rep in range(12):
    for T in (3, 21, 41):
        for td in (8, 12.8, 16, 9.7):
            rows = [fuzz_lines(100000 + 977 * rep + 31 * T + int(td * 10), 300, td, T)]
            cfg = dict(min_length=MIN_LEN, token_distance=td, max_tokens=T, remove_borders=BORDER, max_keylines=-1)
            recs, cu_k, cu_n = eng.prefilter(rows, *HW, remove_borders=BORDER, min_length=MIN_LEN, max_keylines=-1, token_distance=td, max_tokens=T)
            tb = eng.tokenize(recs, cu_k, cu_n, ddc, dsc, token_distance=td, max_tokens=T, want_mat=True)
            want = O.preprocess(synth.array_to_keylines(rows[0]), (1, 1, *HW), dd, ds, cfg, align_corners=False)
            got = {"klines": tb.klines, "length_klines": tb.length, "angles": tb.angles, "sublines": tb.sublines, "pnt_sublines": tb.pnt,
                   "mask_sublines": tb.mask[..., None], "resp_sublines": tb.resp[..., None], "angle_sublines": tb.angle_sub,
                   "score_sublines": tb.score[..., None], "mat_klines2sublines": tb.mat}
            tot += len(rows[0])
            for k in TOK_KEYS:
                h, r = got[k].cpu().numpy(), want[k][0].numpy()
                ok = h.shape == r.shape and (np.abs(h - r).max() <= 1.2e-7 if "angle" in k else np.array_equal(h, r))
                if not ok:
                    bad += 1; print("MISMATCH", rep, T, td, k, h.shape, r.shape)
            worst = max(worst, (tb.desc.cpu() - want["desc_sublines"][0]).abs().max().item())
print("lines", tot, "mismatching tensors", bad, "worst sampled-descriptor error", worst)
