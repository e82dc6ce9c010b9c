"""torch-fp32 restatement of models/lightglue.py (LightGlue.match 447-477, _forward 506-652 and the blocks it
calls) on the CPU path of the reference: fp32 attention, point pruning always on (pruning_keypoint_thresholds
['cpu'] = -1, lightglue.py:351-357), early stopping on.  TEST INFRASTRUCTURE.

Weights: dict name -> torch tensor, keys exactly as in the reference state_dict
(transformers.{i}.self_attn.Wqkv.weight, ..., log_assignment.{i}.final_proj.weight, token_confidence.{i}.token.0.weight,
posenc.Wr.weight, optional input_proj.weight/bias)."""
import math

import torch
import torch.nn.functional as F

N_LAYERS, HEADS = 9, 4


def sample_descriptors(kpts_px, desc_map, s):
    """lightglue.py:24-41.  kpts_px [1,M,2] pixels, desc_map [1,C,h,w] -> [1,C,M] L2-normalised."""
    b, c, h, w = desc_map.shape
    k = kpts_px - s / 2 + 0.5
    k = k / torch.tensor([(w * s - s / 2 - 0.5), (h * s - s / 2 - 0.5)]).to(k)[None]
    k = k * 2 - 1
    d = F.grid_sample(desc_map, k.view(b, 1, -1, 2), mode="bilinear", align_corners=True)
    return F.normalize(d.reshape(b, c, -1), p=2, dim=1)


def normalize_keypoints(kpts):
    """lightglue.py:45-57 with size=None."""
    size = 1 + kpts.max(-2).values - kpts.min(-2).values
    shift = size / 2
    scale = size.max(-1).values / 2
    return (kpts - shift[..., None, :]) / scale[..., None, None]


def rotate_half(x):
    x = x.unflatten(-1, (-1, 2))
    x1, x2 = x.unbind(dim=-1)
    return torch.stack((-x2, x1), dim=-1).flatten(start_dim=-2)


def rope(freqs, t):
    return (t * freqs[0]) + (rotate_half(t) * freqs[1])


def posenc(t, kpts):
    proj = F.linear(kpts, t["posenc.Wr.weight"])                       # lightglue.py:93-96
    emb = torch.stack([torch.cos(proj), torch.sin(proj)], 0).unsqueeze(-3)
    return emb.repeat_interleave(2, dim=-1)


HALF_ATTENTION = False      # True: the reference's cuda branches (lightglue.py:129-134, 226-230): q.half(), k.half(), v.half() through SDPA


def _attention(q, k, v):
    if q.shape[-2] == 0 or k.shape[-2] == 0:
        return q.new_zeros((*q.shape[:-1], v.shape[-1]))
    if HALF_ATTENTION:                                                  # lightglue.py:131-132
        return F.scaled_dot_product_attention(q.half().contiguous(), k.half().contiguous(), v.half().contiguous()).to(q.dtype)
    s = q.shape[-1] ** -0.5
    attn = F.softmax(torch.einsum("...id,...jd->...ij", q, k) * s, -1)
    return torch.einsum("...ij,...jd->...id", attn, v)


def _ffn(t, p, x):
    h = F.linear(x, t[p + ".ffn.0.weight"], t[p + ".ffn.0.bias"])
    h = F.layer_norm(h, (h.shape[-1],), t[p + ".ffn.1.weight"], t[p + ".ffn.1.bias"])
    return F.linear(F.gelu(h), t[p + ".ffn.3.weight"], t[p + ".ffn.3.bias"])


def self_block(t, i, x, enc):                                          # lightglue.py:173-185
    p = "transformers.%d.self_attn" % i
    qkv = F.linear(x, t[p + ".Wqkv.weight"], t[p + ".Wqkv.bias"])
    qkv = qkv.unflatten(-1, (HEADS, -1, 3)).transpose(1, 2)
    q, k, v = qkv[..., 0], qkv[..., 1], qkv[..., 2]
    ctx = _attention(rope(enc, q), rope(enc, k), v)
    msg = F.linear(ctx.transpose(1, 2).flatten(start_dim=-2), t[p + ".out_proj.weight"], t[p + ".out_proj.bias"])
    return x + _ffn(t, p, torch.cat([x, msg], -1))


def cross_block(t, i, x0, x1):                                         # lightglue.py:216-243 (non-flash branch)
    p = "transformers.%d.cross_attn" % i
    lin = lambda n, x: F.linear(x, t[p + "." + n + ".weight"], t[p + "." + n + ".bias"])
    split = lambda z: z.unflatten(-1, (HEADS, -1)).transpose(1, 2)
    qk0, qk1, v0, v1 = split(lin("to_qk", x0)), split(lin("to_qk", x1)), split(lin("to_v", x0)), split(lin("to_v", x1))
    if HALF_ATTENTION:                                                  # lightglue.py:226-230
        m0, m1 = _attention(qk0, qk1, v1), _attention(qk1, qk0, v0)
        m0, m1 = (z.transpose(1, 2).flatten(start_dim=-2) for z in (m0, m1))
        m0, m1 = lin("to_out", m0), lin("to_out", m1)
        return x0 + _ffn(t, p, torch.cat([x0, m0], -1)), x1 + _ffn(t, p, torch.cat([x1, m1], -1))
    scale = qk0.shape[-1] ** -0.5
    qk0, qk1 = qk0 * scale ** 0.5, qk1 * scale ** 0.5
    sim = torch.einsum("bhid, bhjd -> bhij", qk0, qk1)
    attn01 = F.softmax(sim, dim=-1)
    attn10 = F.softmax(sim.transpose(-2, -1).contiguous(), dim=-1)
    m0 = torch.einsum("bhij, bhjd -> bhid", attn01, v1)
    m1 = torch.einsum("bhji, bhjd -> bhid", attn10.transpose(-2, -1), v0)
    m0, m1 = (z.transpose(1, 2).flatten(start_dim=-2) for z in (m0, m1))
    m0, m1 = lin("to_out", m0), lin("to_out", m1)
    return x0 + _ffn(t, p, torch.cat([x0, m0], -1)), x1 + _ffn(t, p, torch.cat([x1, m1], -1))


def confidence_threshold(i):
    return float(min(max(0.8 + 0.1 * math.exp(-4.0 * i / N_LAYERS), 0), 1))


def log_assignment(t, i, d0, d1):                                      # lightglue.py:278-312
    p = "log_assignment.%d" % i
    m0 = F.linear(d0, t[p + ".final_proj.weight"], t[p + ".final_proj.bias"]) / d0.shape[-1] ** 0.25
    m1 = F.linear(d1, t[p + ".final_proj.weight"], t[p + ".final_proj.bias"]) / d0.shape[-1] ** 0.25
    sim = torch.einsum("bmd,bnd->bmn", m0, m1)
    z0 = F.linear(d0, t[p + ".matchability.weight"], t[p + ".matchability.bias"])
    z1 = F.linear(d1, t[p + ".matchability.weight"], t[p + ".matchability.bias"])
    cert = F.logsigmoid(z0) + F.logsigmoid(z1).transpose(1, 2)
    scores = F.log_softmax(sim, 2) + F.log_softmax(sim.transpose(-1, -2).contiguous(), 2).transpose(-1, -2) + cert
    return scores, sim


def forward(t, kpts0, kpts1, desc0, desc1, depth_confidence=0.95, width_confidence=0.99, filter_threshold=0.1,
            pruning_th=-1, trace=None):
    """lightglue.py:506-652.  kpts [1,M,2] pixels, desc [1,M,D].  Returns dict(matches [K,2], scores [K], stop)."""
    b, m, _ = kpts0.shape
    n = kpts1.shape[1]
    k0, k1 = normalize_keypoints(kpts0).clone(), normalize_keypoints(kpts1).clone()
    if "input_proj.weight" in t:
        desc0 = F.linear(desc0, t["input_proj.weight"], t["input_proj.bias"])
        desc1 = F.linear(desc1, t["input_proj.weight"], t["input_proj.bias"])
    e0, e1 = posenc(t, k0), posenc(t, k1)
    ind0, ind1 = torch.arange(m)[None], torch.arange(n)[None]
    tok0 = tok1 = None
    i = 0
    for i in range(N_LAYERS):
        if desc0.shape[1] == 0 or desc1.shape[1] == 0:
            break
        desc0, desc1 = self_block(t, i, desc0, e0), self_block(t, i, desc1, e1)
        desc0, desc1 = cross_block(t, i, desc0, desc1)
        if trace is not None:
            trace.append((desc0.clone(), desc1.clone(), ind0.clone(), ind1.clone()))
        if i == N_LAYERS - 1:
            continue
        if depth_confidence > 0:                                       # lightglue.py:560-563, 670-681
            p = "token_confidence.%d.token.0" % i
            tok0 = torch.sigmoid(F.linear(desc0, t[p + ".weight"], t[p + ".bias"])).squeeze(-1)
            tok1 = torch.sigmoid(F.linear(desc1, t[p + ".weight"], t[p + ".bias"])).squeeze(-1)
            conf = torch.cat([tok0, tok1], -1)
            ratio = 1.0 - (conf < confidence_threshold(i)).float().sum() / (m + n)
            if ratio > depth_confidence:
                break
        if width_confidence > 0:                                       # lightglue.py:564-579, 659-668
            p = "log_assignment.%d.matchability" % i
            for side in (0, 1):
                d, tok = (desc0, tok0) if side == 0 else (desc1, tok1)
                if not d.shape[-2] > pruning_th:
                    continue
                sc = torch.sigmoid(F.linear(d, t[p + ".weight"], t[p + ".bias"])).squeeze(-1)
                keep = sc > (1 - width_confidence)
                if tok is not None:
                    keep |= tok <= confidence_threshold(i)
                kk = torch.where(keep)[1]
                if side == 0:
                    ind0, desc0, e0 = ind0.index_select(1, kk), desc0.index_select(1, kk), e0.index_select(-2, kk)
                else:
                    ind1, desc1, e1 = ind1.index_select(1, kk), desc1.index_select(1, kk), e1.index_select(-2, kk)
    if desc0.shape[1] == 0 or desc1.shape[1] == 0:
        return dict(matches=torch.zeros((0, 2), dtype=torch.long), scores=torch.zeros(0), stop=i + 1)
    scores, _ = log_assignment(t, i, desc0, desc1)
    max0, max1 = scores.max(2), scores.max(1)                          # filter_matches, lightglue.py:315-331
    mi0, mi1 = max0.indices, max1.indices
    mutual0 = torch.arange(mi0.shape[1])[None] == mi1.gather(1, mi0)
    ms0 = torch.where(mutual0, max0.values.exp(), torch.zeros(()))
    valid0 = mutual0 & (ms0 > filter_threshold)
    a = torch.where(valid0[0])[0]
    bidx = mi0[0][valid0[0]]
    return dict(matches=torch.stack([ind0[0, a], ind1[0, bidx]], -1), scores=ms0[0][valid0[0]], stop=i + 1,
                log_scores=scores)


def match(t, pts0, pts1, desc_map_0, desc_map_1, params, desc_scale, **kw):
    """LightGlue.match, lightglue.py:447-477: pts [N,3] normalised (x, y, score); returns matched rows + details."""
    h, w = params["h"], params["w"]
    k0 = pts0[:, :2] * torch.tensor([w - 1, h - 1], dtype=pts0.dtype)
    k1 = pts1[:, :2] * torch.tensor([w - 1, h - 1], dtype=pts0.dtype)
    d0 = sample_descriptors(k0[None], desc_map_0, desc_scale)[0].transpose(-1, -2).contiguous()[None]
    d1 = sample_descriptors(k1[None], desc_map_1, desc_scale)[0].transpose(-1, -2).contiguous()[None]
    out = forward(t, k0[None], k1[None], d0, d1, **kw)
    m = out["matches"]
    return pts0[m[:, 0]], pts1[m[:, 1]], out
