"""CPU oracle for the LineTR hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

This is an independent CPU restatement (NumPy float64 geometry + stock PyTorch-CPU fp32 ops) of the
algorithm in the read-only reference checkout, written from its behaviour, every function citing
the reference file:line it follows.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it; the product package ``linetr_amd`` never does.

Parity status: PINNED.  ``tests/golden/make_golden.py`` imports the real reference in the build
container (cv2 stub, seeded weights) and freezes its inputs/outputs into ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` checks this file against those fixtures (tokeniser bit-exact,
descriptors <= 2e-5, match matrices identical).  The reference itself has no tests or golden
vectors, so those fixtures are the pin.

Stages (reference file:line):
  cv2_to_arrays        models/line_process.py:203-231   change_cv2_T_np
  line_angles          models/line_process.py:28-41     get_angles
  drop_border_lines    models/line_process.py:59-84     remove_borders
  keep_long_lines      models/line_process.py:6-21      filter_by_length
  walk_along           models/line_process.py:43-57     point_on_line
  tokenize             models/line_process.py:100-196   line_tokenizer
  sample_token_desc    models/line_process.py:86-98     sample_descriptors
  preprocess           models/line_transformer.py:251-275
  superpoint_heads     models/superpoint.py:161-167, 190-193   (8(f) row 2: dense score / dense descriptor producer)
  forward              models/line_transformer.py:225-249 (+ :22-183, models/line_attention.py)
  dist_matrix          models/line_process.py:198-201   get_dist_matrix
  subline2keyline      models/line_transformer.py:277-282
  mutual_nn            models/nn_matcher.py:3-31        nn_matcher_distmat
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------------
# a1-a5: host geometry (float64)
# --------------------------------------------------------------------------------------------


def line_angles(kl: np.ndarray) -> np.ndarray:
    """line_process.py:28-41.  theta = arctan2(dx, dy) (x first!), negatives folded by +pi,
    encoded (cos 2theta, sin 2theta).  Empty input returns [] like the reference (:30-32)."""
    if len(kl) == 0:
        return []
    th = np.arctan2(kl[:, 1, 0] - kl[:, 0, 0], kl[:, 1, 1] - kl[:, 0, 1])
    th = np.where(th < 0, th + np.pi, th)
    return np.stack([np.cos(2 * th), np.sin(2 * th)], axis=1)


def cv2_to_arrays(klines_cv) -> dict:
    """line_process.py:203-231.  Endpoint with the smaller x becomes the start point (equal x ->
    swapped, :212-217); length is lineLength * 2**octave as reported by the detector (:220)."""
    sp, ep, ln = [], [], []
    for l in klines_cv:
        a = [l.startPointX, l.startPointY]
        b = [l.endPointX, l.endPointY]
        if not (a[0] < b[0]):
            a, b = b, a
        sp.append(a)
        ep.append(b)
        ln.append(l.lineLength * (2 ** l.octave))
    sp = np.asarray(sp)
    ep = np.asarray(ep)
    kl = np.stack((sp, ep), axis=1)  # raises on zero lines exactly like the reference (quirk 23)
    return {"klines": kl, "length_klines": np.asarray(ln), "angles": line_angles(kl)}


def drop_border_lines(lines: dict, border, height, width, valid_mask=None) -> dict:
    """line_process.py:59-84.  Keep lines with both endpoints in [b, W-b) x [b, H-b); clip all
    coordinates IN PLACE to W-eps-b / H-eps-b (eps=1e-3); an ndarray valid mask keeps a line if
    either floor(endpoint) is valid; a non-ndarray mask (torch tensor) is ignored (:76)."""
    kl = lines["klines"]
    ok = np.ones(len(kl), dtype=bool)
    for e in (0, 1):
        ok &= (kl[:, e, 0] >= border) & (kl[:, e, 0] < width - border)
        ok &= (kl[:, e, 1] >= border) & (kl[:, e, 1] < height - border)
    eps = 0.001
    kl[:, :, 0] = kl[:, :, 0].clip(max=width - eps - border)
    kl[:, :, 1] = kl[:, :, 1].clip(max=height - eps - border)
    if isinstance(valid_mask, np.ndarray):
        s = np.floor(kl[:, 0]).astype(int)
        e = np.floor(kl[:, 1]).astype(int)
        ok &= (valid_mask[s[:, 1], s[:, 0]] + valid_mask[e[:, 1], e[:, 0]]).astype(bool)
    return {k: v[ok] for k, v in lines.items()}


def keep_long_lines(lines: dict, min_length, max_lines) -> dict:
    """line_process.py:6-21.  length > min_length (strict); descending order via reversed
    np.argsort; slice [:max_lines] (max_lines=-1 drops the shortest survivor); recompute angles."""
    ln = lines["length_klines"]
    sel = ln > min_length
    kl, ln = lines["klines"][sel], ln[sel]
    order = np.argsort(ln)[::-1][:max_lines]
    kl, ln = kl[order], ln[order]
    return {"klines": kl, "length_klines": ln, "angles": line_angles(kl)}


def walk_along(sp: np.ndarray, ep: np.ndarray, dist: np.ndarray) -> np.ndarray:
    """line_process.py:43-57 for a vector of arclengths: slope form x = sqrt(d^2/(1+m^2)), y = m x
    (same float64 operation order), vertical special case; asserts 0 <= d <= |line| (:44-45)."""
    geo = np.sqrt(np.sum((ep - sp) ** 2))
    assert np.all(dist >= 0), "distance should be positive!"
    assert np.all(geo >= dist), "distance should be smaller than line length!"
    vx, vy = ep[0] - sp[0], ep[1] - sp[1]
    if vx != 0:
        m = vy / vx
        x = np.sqrt(dist * dist / (1 + m * m))
        y = m * x
    else:
        x = np.zeros_like(dist)
        y = dist if vy > 0 else -dist
    return np.stack([x + sp[0], y + sp[1]], axis=1)


def sample_token_desc(tokens: torch.Tensor, dense_descriptor: torch.Tensor, s: int = 8,
                      align_corners: bool | None = None) -> torch.Tensor:
    """line_process.py:86-98.  tokens [1,N,T,2] -> [1,256,N*T] L2-normalised bilinear samples.
    align_corners follows the reference's torch-version switch (:93) unless forced."""
    b, c, h, w = dense_descriptor.shape
    kp = tokens - s / 2 + 0.5
    kp = kp / torch.tensor([(w * s - s / 2 - 0.5), (h * s - s / 2 - 0.5)]).to(kp)[None]
    kp = kp * 2 - 1
    if align_corners is None:
        align_corners = int(torch.__version__[2]) > 2
    args = {"align_corners": True} if align_corners else {}
    d = F.grid_sample(dense_descriptor, kp.view(b, 1, -1, 2), mode="bilinear", **args)
    return F.normalize(d.reshape(b, c, -1), p=2, dim=1)


def tokenize(lines: dict, token_distance, max_tokens: int, dense_descriptor: torch.Tensor,
             dense_score: torch.Tensor, image_hw, align_corners: bool | None = None) -> dict:
    """line_process.py:100-196.  Mutates ``lines`` (and the end points inside lines['klines'])
    exactly like the reference and returns the same dict with the 11 tensor entries."""
    height, width = image_hw
    kl_all, len_all, ang_all = lines["klines"], lines["length_klines"], lines["angles"]
    T = max_tokens
    subl, ntok_sub, pnts, masks, resp, ang, n_sub_per_line = [], [], [], [], [], [], []
    for i in range(len(kl_all)):
        kl = kl_all[i]
        n_tok = int(math.ceil(len_all[i] / token_distance))                       # :109
        d = np.arange(max(n_tok - 1, 0), dtype=np.float64) * token_distance       # :110-113
        toks = walk_along(kl[0].copy(), kl[1].copy(), d) if len(d) else np.zeros((0, 2))
        kl[1, 0] = min(kl[1, 0], width - 0.6)                                     # :114-116 (view!)
        kl[1, 1] = min(kl[1, 1], height - 0.6)
        toks = np.concatenate([toks, kl[1][None]], axis=0)                        # :117
        n_sub = int(math.ceil(n_tok / T))                                         # :121
        sl = np.zeros((n_sub, 2, 2))
        sl[0, 0] = kl[0]
        sl[-1, 1] = kl[1]
        for j in range(n_sub - 1):                                                # :125-128
            mid = toks[(j + 1) * T - 1]
            sl[j, 1] = mid
            sl[j + 1, 0] = mid
        p = np.zeros((n_sub, T, 2))
        mk = np.zeros((n_sub, T + 1, 1))
        mk[:, 0] = 1                                                              # :135
        for j in range(n_sub):                                                    # :136-141
            part = toks[j * T:(j + 1) * T]
            p[j, :len(part)] = part
            mk[j, 1:len(part) + 1] = 1
        geo = np.sqrt(((sl[:, 1] - sl[:, 0]) ** 2).sum(-1))                       # :148-149
        subl.append(sl)
        pnts.append(p)
        masks.append(mk)
        resp.append((geo / (token_distance * T))[:, None])
        ang.append(np.repeat(ang_all[i][None], n_sub, axis=0))                    # :151
        n_sub_per_line.append(n_sub)
    dev = dense_descriptor.device
    f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float().to(dev)     # :154-160
    sublines = f32(np.concatenate(subl).reshape(-1, 2, 2))
    tokens = f32(np.concatenate(pnts).reshape(-1, T, 2))
    mask = f32(np.concatenate(masks).reshape(-1, T + 1, 1))
    responses = f32(np.concatenate(resp).reshape(-1, 1))
    angles = f32(np.concatenate(ang).reshape(-1, 2))
    N = sublines.shape[0]
    A = torch.zeros((len(kl_all), N), device=dev)                                 # :163-167
    c = 0
    for i, n in enumerate(n_sub_per_line):
        A[i, c:c + n] = 1 / n
        c += n
    desc = sample_token_desc(tokens[None], dense_descriptor, 8, align_corners)[0]  # :171-172
    desc = desc.reshape(dense_descriptor.shape[1], N, T).permute(1, 2, 0)
    sc = dense_score.transpose(1, 2)                                              # :174-179
    ij = torch.round(tokens).long().reshape(-1, 2)
    ij[:, 0] = ij[:, 0].clip(max=sc.shape[1] - 1)
    ij[:, 1] = ij[:, 1].clip(max=sc.shape[2] - 1)
    score = sc[0][ij[:, 0], ij[:, 1]].reshape(N, T, 1)
    lines["klines"] = f32(kl_all)[None]                                           # :182-184
    lines["length_klines"] = f32(len_all)[None]
    lines["angles"] = f32(ang_all)[None]
    lines["sublines"] = sublines[None]
    lines["pnt_sublines"] = tokens[None]
    lines["mask_sublines"] = mask[None]
    lines["resp_sublines"] = responses[None]
    lines["angle_sublines"] = angles[None]
    lines["desc_sublines"] = desc[None]
    lines["score_sublines"] = score[None]
    lines["mat_klines2sublines"] = A[None]
    return lines


def preprocess(klines_cv, image_shape, dense_descriptor, dense_score, config: dict,
               valid_mask=None, align_corners: bool | None = None) -> dict:
    """line_transformer.py:251-275 (a9).  image_shape is the 4-tuple (1,1,H,W)."""
    lines = cv2_to_arrays(klines_cv)
    _, _, height, width = image_shape
    if valid_mask is None:
        valid_mask = np.ones((height, width))
    lines = drop_border_lines(lines, config["remove_borders"], height, width, valid_mask)
    lines = keep_long_lines(lines, config["min_length"], config["max_keylines"])
    if len(lines["klines"]) == 0:
        return lines
    return tokenize(lines, config["token_distance"], config["max_tokens"], dense_descriptor,
                    dense_score, (height, width), align_corners)


# --------------------------------------------------------------------------------------------
# a10-a18: model forward, as executed by the reference (full S-token descriptive layer)
# --------------------------------------------------------------------------------------------


def _mlp(sd: dict, prefix: str, x: torch.Tensor) -> torch.Tensor:
    """line_transformer.py:9-20 on [rows, C] (Conv1d k=1 == per-row Linear); eval-mode BN."""
    idx = 0
    while f"{prefix}.{idx}.weight" in sd:
        w = sd[f"{prefix}.{idx}.weight"]
        x = F.linear(x, w[:, :, 0], sd[f"{prefix}.{idx}.bias"])
        idx += 1
        if f"{prefix}.{idx}.running_mean" in sd:
            x = F.batch_norm(x, sd[f"{prefix}.{idx}.running_mean"], sd[f"{prefix}.{idx}.running_var"],
                             sd[f"{prefix}.{idx}.weight"], sd[f"{prefix}.{idx}.bias"], False, 0.0, 1e-5)
            x = F.relu(x)
            idx += 2
    return x


def default_ret() -> dict:
    """line_transformer.py:284-291."""
    return {"klines": torch.empty((1, 0, 2, 2)), "sublines": torch.empty((1, 0, 2, 2)),
            "line_desc": torch.empty((1, 256, 0)), "mat_klines2sublines": torch.empty((1, 0, 0))}


def forward(sd: dict, data: dict, image_shape=(480, 640), n_heads: int = 4) -> dict:
    """line_transformer.py:225-249.  ``sd`` = state_dict of torch CPU tensors; ``image_shape`` is
    the CONSTRUCTOR-time shape used by normalize_keylines (quirk 19)."""
    if len(data["klines"]) == 0:
        return default_ret()
    sub = data["sublines"][0]            # [N,2,2]
    pnt = data["pnt_sublines"][0]        # [N,T,2]
    desc = data["desc_sublines"][0]      # [N,T,256]
    score = data["score_sublines"][0]    # [N,T,1]
    mask = data["mask_sublines"][0]      # [N,S,1]
    resp = data["resp_sublines"][0]      # [N,1]
    ang = data["angle_sublines"][0]      # [N,2]
    N, T, d = desc.shape
    height, width = image_shape[-2:]
    ctr = torch.tensor([width / 2.0, height / 2.0])                               # :22-38
    scale = torch.tensor(float(max(width, height))) * 0.7
    sub_n = (sub - ctr) / scale
    pnt_n = (pnt - ctr) / scale
    # positional encoders :40-73
    mid = (sub_n[:, 0] + sub_n[:, 1]) / 2.0
    line_pos = _mlp(sd, "klenc.line_position_enc.encoder", torch.cat([mid, resp, ang], dim=1))  # [N,256]
    word_pos = _mlp(sd, "klenc.word_position_enc.encoder",
                    torch.cat([pnt_n, score], dim=-1).reshape(N * T, 3)).reshape(N, T, d)
    x = desc + word_pos                                                            # :117
    x = torch.cat([sd["klenc.cls_token"].reshape(1, 1, d).expand(N, 1, d), x], dim=1)  # :120-121
    n_layers = 0
    while f"klenc.desc_layers.{n_layers}.slf_attn.fc.weight" in sd:
        n_layers += 1
    p = f"klenc.desc_layers.{n_layers - 1}"   # :123-125 -- every layer sees the same input; last wins
    S = T + 1
    dh = d // n_heads
    q = F.linear(x, sd[f"{p}.slf_attn.w_qs.weight"], sd[f"{p}.slf_attn.w_qs.bias"]).view(N, S, n_heads, dh).transpose(1, 2)
    k = F.linear(x, sd[f"{p}.slf_attn.w_ks.weight"], sd[f"{p}.slf_attn.w_ks.bias"]).view(N, S, n_heads, dh).transpose(1, 2)
    v = F.linear(x, sd[f"{p}.slf_attn.w_vs.weight"], sd[f"{p}.slf_attn.w_vs.bias"]).view(N, S, n_heads, dh).transpose(1, 2)
    att = torch.matmul(q / (dh ** 0.5), k.transpose(2, 3))                        # line_attention.py:14
    att = att.masked_fill(mask.view(N, 1, S, 1) == 0, -1e9)                       # :16 (query rows!)
    att = F.softmax(att, dim=-1)
    o = torch.matmul(att, v).transpose(1, 2).reshape(N, S, d)
    o = F.linear(o, sd[f"{p}.slf_attn.fc.weight"], sd[f"{p}.slf_attn.fc.bias"]) + x
    o = F.layer_norm(o, (d,), sd[f"{p}.slf_attn.layer_norm.weight"], sd[f"{p}.slf_attn.layer_norm.bias"], 1e-6)
    f = F.linear(F.gelu(F.linear(o, sd[f"{p}.pos_ffn.w_1.weight"], sd[f"{p}.pos_ffn.w_1.bias"])),
                 sd[f"{p}.pos_ffn.w_2.weight"], sd[f"{p}.pos_ffn.w_2.bias"]) + o   # line_attention.py:86-94
    f = F.layer_norm(f, (d,), sd[f"{p}.pos_ffn.layer_norm.weight"], sd[f"{p}.pos_ffn.layer_norm.bias"], 1e-6)
    z = line_pos + f[:, 0, :]                                                      # :128   [N,256]
    l = 0
    while f"selfattn.layers.{l}.attn.merge.weight" in sd:                          # :168-183
        a = f"selfattn.layers.{l}.attn"
        qkv = [F.linear(z, sd[f"{a}.proj.{j}.weight"][:, :, 0], sd[f"{a}.proj.{j}.bias"]).view(N, dh, n_heads)
               for j in range(3)]                                                  # channel c = d*4+h (:151)
        sc = torch.einsum("ndh,mdh->hnm", qkv[0], qkv[1]) / dh ** 0.5              # :134
        pr = F.softmax(sc, dim=-1)
        msg = torch.einsum("hnm,mdh->ndh", pr, qkv[2]).reshape(N, d)
        msg = F.linear(msg, sd[f"{a}.merge.weight"][:, :, 0], sd[f"{a}.merge.bias"])
        z = z + _mlp(sd, f"selfattn.layers.{l}.mlp", torch.cat([z, msg], dim=1))   # :166,:181
        l += 1
    z = F.linear(z, sd["final_proj.weight"][:, :, 0], sd["final_proj.bias"])      # :245
    z = F.normalize(z, p=2, dim=1)                                                 # :246
    data["line_desc"] = z.t()[None].contiguous()                                   # [1,256,N]
    return data


def forward_batch(sd: dict, data: dict, image_shape=(480, 640), n_heads: int = 4) -> dict:
    """The same forward on a dict whose tensors carry a batch axis B >= 1 with a fixed number of sub-lines per image --
    the way train.py:163-164 calls the model on the dataset builder's fixed-size samples (util_lines.py:670-766).
    Images are independent (the signature attention runs per batch element, line_transformer.py:132-154), so this is
    forward() image by image; line_desc comes back as [B,256,N]."""
    B = data["sublines"].shape[0]
    keys = ("sublines", "pnt_sublines", "desc_sublines", "score_sublines", "mask_sublines", "resp_sublines", "angle_sublines")
    descs = []
    for b in range(B):
        one = {k: data[k][b:b + 1] for k in keys}
        one["klines"] = data["klines"][b:b + 1]
        descs.append(forward(sd, one, image_shape, n_heads)["line_desc"])
    data["line_desc"] = torch.cat(descs, dim=0)
    return data


def _mlp_train(sd: dict, prefix: str, x: torch.Tensor, momentum: float) -> torch.Tensor:
    """_mlp with BatchNorm1d in TRAINING mode (train.py:127 -> model.train()): batch statistics over all rows, and the running
    statistics / num_batches_tracked entries of ``sd`` updated IN PLACE the way torch.nn.BatchNorm1d does."""
    idx = 0
    while f"{prefix}.{idx}.weight" in sd:
        x = F.linear(x, sd[f"{prefix}.{idx}.weight"][:, :, 0], sd[f"{prefix}.{idx}.bias"])
        idx += 1
        if f"{prefix}.{idx}.running_mean" in sd:
            x = F.batch_norm(x, sd[f"{prefix}.{idx}.running_mean"], sd[f"{prefix}.{idx}.running_var"],
                             sd[f"{prefix}.{idx}.weight"], sd[f"{prefix}.{idx}.bias"], True, momentum, 1e-5)
            if f"{prefix}.{idx}.num_batches_tracked" in sd:
                sd[f"{prefix}.{idx}.num_batches_tracked"] += 1
            x = F.relu(x)
            idx += 2
    return x


def forward_train(sd: dict, data: dict, image_shape=(480, 640), n_heads: int = 4, momentum: float = 0.1) -> dict:
    """The TRAIN-MODE forward (8(f) row 4): models/line_transformer.py:225-249 with the module in .train() (train.py:127), called
    on a batch [B,N,...] of fixed-size samples (train.py:163-164).  Every BatchNorm1d of the three MLP stacks normalises with the
    statistics of the whole batch -- B*N*T token positions in the word encoder ([B*N,3,T] input, :63-68), B*N sub-lines in the line
    encoder and the signature MLPs -- and ``sd``'s running statistics are updated in place.  Dropout is taken at probability 0
    (models/line_attention.py:11,39,84; a fixture cannot hold an RNG stream).  line_desc comes back as [B,256,N]."""
    sub, pnt, desc = data["sublines"], data["pnt_sublines"], data["desc_sublines"]
    B, N, T, d = desc.shape
    score, mask = data["score_sublines"], data["mask_sublines"]
    resp, ang = data["resp_sublines"], data["angle_sublines"]
    height, width = image_shape[-2:]
    ctr = torch.tensor([width / 2.0, height / 2.0])
    scale = torch.tensor(float(max(width, height))) * 0.7
    sub_n = ((sub - ctr) / scale).reshape(B * N, 2, 2)
    pnt_n = ((pnt - ctr) / scale).reshape(B * N, T, 2)
    mid = (sub_n[:, 0] + sub_n[:, 1]) / 2.0
    line_pos = _mlp_train(sd, "klenc.line_position_enc.encoder",
                          torch.cat([mid, resp.reshape(B * N, 1), ang.reshape(B * N, 2)], dim=1), momentum)
    word_pos = _mlp_train(sd, "klenc.word_position_enc.encoder",
                          torch.cat([pnt_n, score.reshape(B * N, T, 1)], dim=-1).reshape(B * N * T, 3), momentum).reshape(B * N, T, d)
    x = desc.reshape(B * N, T, d) + word_pos
    M = B * N
    x = torch.cat([sd["klenc.cls_token"].reshape(1, 1, d).expand(M, 1, d), x], dim=1)
    n_layers = 0
    while f"klenc.desc_layers.{n_layers}.slf_attn.fc.weight" in sd:
        n_layers += 1
    p = f"klenc.desc_layers.{n_layers - 1}"
    S, dh = T + 1, d // n_heads
    q = F.linear(x, sd[f"{p}.slf_attn.w_qs.weight"], sd[f"{p}.slf_attn.w_qs.bias"]).view(M, S, n_heads, dh).transpose(1, 2)
    k = F.linear(x, sd[f"{p}.slf_attn.w_ks.weight"], sd[f"{p}.slf_attn.w_ks.bias"]).view(M, S, n_heads, dh).transpose(1, 2)
    v = F.linear(x, sd[f"{p}.slf_attn.w_vs.weight"], sd[f"{p}.slf_attn.w_vs.bias"]).view(M, S, n_heads, dh).transpose(1, 2)
    att = torch.matmul(q / (dh ** 0.5), k.transpose(2, 3))
    att = att.masked_fill(mask.reshape(M, 1, S, 1) == 0, -1e9)
    att = F.softmax(att, dim=-1)
    o = torch.matmul(att, v).transpose(1, 2).reshape(M, S, d)
    o = F.linear(o, sd[f"{p}.slf_attn.fc.weight"], sd[f"{p}.slf_attn.fc.bias"]) + x
    o = F.layer_norm(o, (d,), sd[f"{p}.slf_attn.layer_norm.weight"], sd[f"{p}.slf_attn.layer_norm.bias"], 1e-6)
    f = F.linear(F.gelu(F.linear(o, sd[f"{p}.pos_ffn.w_1.weight"], sd[f"{p}.pos_ffn.w_1.bias"])),
                 sd[f"{p}.pos_ffn.w_2.weight"], sd[f"{p}.pos_ffn.w_2.bias"]) + o
    f = F.layer_norm(f, (d,), sd[f"{p}.pos_ffn.layer_norm.weight"], sd[f"{p}.pos_ffn.layer_norm.bias"], 1e-6)
    z = line_pos + f[:, 0, :]                                                      # [B*N,256]
    l = 0
    while f"selfattn.layers.{l}.attn.merge.weight" in sd:
        a = f"selfattn.layers.{l}.attn"
        qkv = [F.linear(z, sd[f"{a}.proj.{j}.weight"][:, :, 0], sd[f"{a}.proj.{j}.bias"]).view(B, N, dh, n_heads) for j in range(3)]
        sc = torch.einsum("bndh,bmdh->bhnm", qkv[0], qkv[1]) / dh ** 0.5           # per image (:132-136)
        pr = F.softmax(sc, dim=-1)
        msg = torch.einsum("bhnm,bmdh->bndh", pr, qkv[2]).reshape(M, d)
        msg = F.linear(msg, sd[f"{a}.merge.weight"][:, :, 0], sd[f"{a}.merge.bias"])
        z = z + _mlp_train(sd, f"selfattn.layers.{l}.mlp", torch.cat([z, msg], dim=1), momentum)
        l += 1
    z = F.linear(z, sd["final_proj.weight"][:, :, 0], sd["final_proj.bias"])
    z = F.normalize(z, p=2, dim=1)
    data["line_desc"] = z.reshape(B, N, d).transpose(1, 2).contiguous()
    return data


# --------------------------------------------------------------------------------------------
# a19-a21: matcher (NumPy, like the reference)
# --------------------------------------------------------------------------------------------


def dist_matrix(desc0: np.ndarray, desc1: np.ndarray) -> np.ndarray:
    """line_process.py:198-201.  [b,256,N0],[b,256,N1] -> clip(2 - 2 d0^T d1, 0) [b,N0,N1]."""
    return (2.0 - 2.0 * np.einsum("bdn,bdm->bnm", desc0, desc1)).clip(min=0)


def subline2keyline(dist_sub: np.ndarray, A0, A1) -> np.ndarray:
    """line_transformer.py:277-282.  Mean sub-line distance per key-line pair, [1,K0,K1]."""
    A0 = A0.cpu().numpy() if torch.is_tensor(A0) else np.asarray(A0)
    A1 = A1.cpu().numpy() if torch.is_tensor(A1) else np.asarray(A1)
    return (A0 @ dist_sub @ A1.T)[None]


def mutual_nn(dist: np.ndarray, thr, mutual: bool = True) -> np.ndarray:
    """nn_matcher.py:3-31.  [1,n0,n1] -> float64 0/1 matrix; first-index argmin, strict <,
    optional mutual check; zeros if either side is empty."""
    n0, n1 = dist.shape[1], dist.shape[2]
    out = np.zeros((1, n0, n1))
    if n0 == 0 or n1 == 0:
        return out
    dm = dist[0].clip(min=0)
    j = np.argmin(dm, axis=1)
    best = dm[np.arange(n0), j]
    keep = best < thr
    if mutual:
        i_back = np.argmin(dm, axis=0)
        keep &= np.arange(n0) == i_back[j]
    out[0, np.arange(n0)[keep], j[keep]] = 1
    return out


def match_lines(desc0, desc1, A0, A1, thr):
    """The line branch of Matching.forward, models/matching.py:77-84.  Returns (matches, Dk)."""
    d0 = desc0.cpu().numpy() if torch.is_tensor(desc0) else desc0
    d1 = desc1.cpu().numpy() if torch.is_tensor(desc1) else desc1
    D = dist_matrix(d0, d1)[0]
    Dk = subline2keyline(D, A0, A1)
    return mutual_nn(Dk, thr, True), Dk


def point_nn(desc0: np.ndarray, desc1: np.ndarray, thr=0.8, mutual=True):
    """nn_matcher.py:33-42 (point matcher, the section-8(f) 'next' row)."""
    dm = (2.0 - 2.0 * (desc0.T @ desc1)).clip(min=0)[None]
    return mutual_nn(dm, thr, mutual), dm


# ---- SuperPoint head post-processing -------------------------------------

def superpoint_heads(score_logits, desc_raw):
    """models/superpoint.py:161-167 and :190-193, stock PyTorch-CPU fp32 ops.

    score_logits [B,65,Hc,Wc] (convPb output) -> dense_score [B,8Hc,8Wc]: softmax over the 65 channels, dustbin
    (last channel) dropped, channel c = 8*dy + dx moved to pixel (8h+dy, 8w+dx).
    desc_raw [B,256,Hc,Wc] (convDb output) -> dense_descriptor [B,256,Hc,Wc] = x / max(||x||_2, 1e-12) over channels.
    """
    import torch
    sl = torch.as_tensor(np.asarray(score_logits, dtype=np.float32))
    dr = torch.as_tensor(np.asarray(desc_raw, dtype=np.float32))
    p = torch.softmax(sl, 1)[:, :-1]
    b, _, h, w = p.shape
    p = p.permute(0, 2, 3, 1).reshape(b, h, w, 8, 8).permute(0, 1, 3, 2, 4).reshape(b, h * 8, w * 8)
    d = torch.nn.functional.normalize(dr, p=2, dim=1)
    return p.numpy(), d.numpy()
